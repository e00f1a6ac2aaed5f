"""The three callers (VAE / IWAE / BNN counterparts in zhusuan-pytorch_amd/examples) against golden values
captured from the reference's example models with identical weights, data and epsilon draws
(SURVEY.md section 8a rows 12-14).  Parity bar of BASELINE.json: 1e-4 relative on the ELBO.
"host" back-end on CPU (C oracle injected) and, marked gpu, the HIP library.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers as H
import host_backend
import zhusuan as zs
from examples import vae_mnist, iwae, bnn_vi


def T(a, dev):
    return torch.tensor(np.asarray(a, dtype=np.float32), device=dev)


def rel(a, b):
    a = float(a.detach()) if isinstance(a, torch.Tensor) else float(a)
    b = float(b)
    return abs(a - b) / max(abs(b), 1e-30)


def _check_z(z, ref, dev):
    """z = mean + std*eps is bit-exact given identical (mean, std); end to end the encoder GEMMs run in
    hipBLASLt on the GPU (other summation order than the reference's MKL), so there it is close, not equal."""
    z = z.detach().cpu().numpy()
    if dev.type == "cpu":
        assert np.array_equal(z, ref)
    else:
        np.testing.assert_allclose(z, ref, rtol=2e-5, atol=2e-5)


GRAD_STRIDE, GRAD_STRIDE_BIG = 97, 997      # tests/golden/gen_golden.py: sampling strides of the stored gradients


def _check_grads(g, model, rtol_norm=5e-4, atol_scale=1e-4, full=False):
    """Gradient gate of SURVEY.md section 8d: <= 1e-3 relative, element by element wherever the fixture holds the
    gradient itself -- whole tensors up to 32768 elements and every 97th element of the large weight matrices in the
    small goldens, every 997th element of every gradient in the config-shape goldens -- plus per-parameter norms.

    Where the fixture also holds the SAME reference code evaluated in float64 ("...64" arrays: the IWAE goldens), the
    target is that float64 result and the absolute allowance is what the reference's own fp32 run needs against it
    (VIMCO's learning signal subtracts two ~|log w| = 550-sized fp32 numbers, SURVEY.md 7.4-6: the reference's fp32
    gradients are themselves only 1e-4 .. 1e-3-accurate): |got - ref64| <= 1e-3 |ref64| + 1.5 max|ref32 - ref64|.
    Without a float64 run: |got - ref32| <= 1e-3 |ref32| + atol_scale * max|ref32| (an element of a weight gradient is
    a sum of hundreds of products of both signs; its rounding error scales with the tensor, not with the element)."""
    names = [str(n) for n in g["grad_names"]]
    named = list(model.named_parameters())
    assert names == [n for n, _ in named]
    norms = np.array([float(p.grad.double().norm()) for _, p in named])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=rtol_norm, atol=1e-6)
    n_elementwise = 0
    for n, p in named:
        got = p.grad.reshape(-1).cpu().numpy().astype(np.float64)
        # the tensor's size and the reference's own fp32 error, from the largest sample the fixture holds of it
        big = [k for k in ("gfull", "gstride", "ghead") if "%s_%s" % (k, n) in g.files][0]
        has64 = "%s64_%s" % (big, n) in g.files
        if has64:
            scale = max(float(g["grad_absmax64"][names.index(n)]), 1e-30)
            ref_err = float(np.abs(g["%s_%s" % (big, n)].astype(np.float64) - g["%s64_%s" % (big, n)]).max())
        for kind, stride in (("ghead", None), ("gfull", 1), ("gstride", GRAD_STRIDE if full else GRAD_STRIDE_BIG)):
            key = "%s_%s" % (kind, n)
            if key not in g.files:
                continue
            ref32 = g[key].astype(np.float64)
            have = got[:8] if stride is None else got[::stride]
            if has64:
                np.testing.assert_allclose(have, g["%s64_%s" % (kind, n)], rtol=1e-3, atol=1.5 * ref_err + atol_scale * scale,
                                           err_msg=key + " (float64 reference)")
            else:
                np.testing.assert_allclose(have, ref32, rtol=1e-3, atol=atol_scale * max(float(np.abs(got).max()), 1e-30),
                                           err_msg=key)
            if stride is not None:
                n_elementwise += ref32.size
    return n_elementwise


@pytest.mark.parametrize("tag,B", [("small", 8), ("c1", 64), ("c2", 512)])
def test_vae(dev, tag, B):
    g = load_golden("g_vae_" + tag)
    model = vae_mnist.build(batch_size=B, device=dev)
    H.load_params_into(model, 1000 + B)
    x, e1, e2 = H.vae_data(B)
    with zs.inject_epsilon([e1, e2]):            # two draws per step, the second one is used
        loss = model({"x": T(x, dev)})
    assert loss.dim() == 0
    assert rel(loss, g["loss"]) < 2e-5
    gen, var = model.generator, model.variational
    assert rel(gen.nodes["z"].log_prob(), g["logpz"]) < 2e-5
    assert rel(gen.nodes["x"].log_prob(), g["logpx"]) < 2e-5
    assert rel(var.nodes["z"].log_prob(), g["logqz"]) < 2e-5
    model.zero_grad()
    loss.backward()
    n_el = _check_grads(g, model, full=(tag == "small"))
    assert (n_el > 60000) == (tag == "small")         # the small golden pins the gradients element by element
    if tag == "small":
        _check_z(var.nodes["z"].dist.sample_cache, g["z"], dev)
        np.testing.assert_allclose(gen.cache["x_mean"].detach().cpu().numpy(), g["x_mean"], rtol=1e-4, atol=1e-6)
    # generation path (vae_mnist.py:129-136): fresh prior sample, Bernoulli draw
    gen({})
    assert gen.cache["x_mean"].shape == (B, 784)
    assert set(np.unique(gen.nodes["x"].tensor.cpu().numpy())) <= {0.0, 1.0}


@pytest.mark.parametrize("est", ["sgvb", "vimco"])
@pytest.mark.parametrize("tag,B,K,hidden", [("small", 8, 5, 32), ("c3", 256, 50, 500), ("c4g", 2048, 50, 500)])
@pytest.mark.parametrize("fused_logits", [False, True])
def test_iwae(dev, est, tag, B, K, hidden, fused_logits):
    """small: every tensor; c3 (BASELINE config 3 = per-GPU shape of config 4): every log-importance-weight and its three
    terms; c4g: the GLOBAL batch of config 4 (2048 x 50) in one process, scalars + gradient statistics + slices (both estimators, both
    Bernoulli paths)."""
    if tag == "c4g" and dev.type == "cpu":
        pytest.skip("103 M-element Bernoulli stream through the serial C oracle: GPU only")
    g = load_golden("g_iwae_%s_%s" % (est, tag))
    model = iwae.build(n_samples=K, estimator=est, hidden=hidden, device=dev, fused_logits=fused_logits)
    H.load_params_into(model, 2000 + B + K)
    x, e1, e2 = H.iwae_data(B, K)
    with zs.inject_epsilon([e1, e2]):
        loss = model({"x": T(x, dev)})
    assert rel(loss, g["loss"]) < 5e-5, (float(loss), float(g["loss"]))
    assert rel(model.last_iw_bound.mean(), g["iw_bound"]) < 2e-5
    model.zero_grad()
    loss.backward()
    n_el = _check_grads(g, model, rtol_norm=1e-3, full=(tag == "small"))
    assert n_el > (50000 if tag == "small" else 1000 if tag == "c3" else -1)
    if "loss64" in g.files:       # the objective itself against the float64 evaluation of the reference code
        ref_err = abs(float(g["loss"]) - float(g["loss64"]))
        assert abs(float(loss) - float(g["loss64"])) <= 2e-5 * abs(float(g["loss64"])) + 1.5 * ref_err
    gen, var = model.generator, model.variational
    lq = var.nodes["z"].log_prob()
    assert tuple(lq.shape) == (K, B) and lq.stride() == (1, K)      # K-fastest rows, reference shape
    lpz, lpx = gen.nodes["z"].log_prob(), gen.nodes["x"].log_prob()
    log_w = (lpz + lpx - lq).detach().cpu().numpy()
    if tag == "small":
        _check_z(var.nodes["z"].dist.sample_cache, g["z"], dev)
    if tag != "c4g":                         # all K*B log-importance-weights and their three terms
        np.testing.assert_allclose(log_w, g["log_w"], rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(lq.detach().cpu().numpy(), g["logqz"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(lpz.detach().cpu().numpy(), g["logpz"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(lpx.detach().cpu().numpy(), g["logpx"], rtol=2e-5, atol=2e-4)
    else:
        np.testing.assert_allclose(log_w[:, 0], g["log_w_col0"], rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(log_w[0, ::16], g["log_w_row0_every16"], rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(model.last_iw_bound.detach().cpu().numpy()[::16], g["bound_b_every16"], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("est", ["sgvb", "vimco"])
@pytest.mark.parametrize("tag,B,K,hidden", [("small", 8, 5, 32), ("c3", 256, 50, 500)])
@pytest.mark.parametrize("fused_logits,paired", [(False, True), (True, True), (False, False)])
def test_iwae_on_the_in_kernel_philox_stream(dev, est, tag, B, K, hidden, fused_logits, paired):
    """The PRODUCTION sampling path against the reference in one hop (nuance (i)): nothing is injected here.
    The fixtures hold the reference's outputs for draws taken from the Philox4x32-10 stream of (seed, call ids 0 and 1)
    (tests/golden/gen_golden.py:gen_iwae_philox); the package, seeded the same way, draws inside its sampling kernel --
    both draws of the latent in ONE launch (`paired`), or one launch per draw -- and must land on the same objective,
    log-importance-weights and gradients as the reference."""
    g = load_golden("g_iwae_%s_%s_philox" % (est, tag))
    seed = int(g["philox_seed"])
    model = iwae.build(n_samples=K, estimator=est, hidden=hidden, device=dev, fused_logits=fused_logits)
    H.load_params_into(model, 2000 + B + K)
    x, _, _ = H.iwae_data(B, K)
    _seed_philox(dev, seed)
    with zs.pair_draws(paired):
        loss = model({"x": T(x, dev)})
    assert rel(loss, g["loss"]) < 5e-5, (float(loss), float(g["loss"]))
    assert rel(model.last_iw_bound.mean(), g["iw_bound"]) < 2e-5
    model.zero_grad()
    loss.backward()
    n_el = _check_grads(g, model, rtol_norm=1e-3, full=(tag == "small"))
    assert n_el > (50000 if tag == "small" else 1000)
    ref_err = abs(float(g["loss"]) - float(g["loss64"]))
    assert abs(float(loss.detach()) - float(g["loss64"])) <= 2e-5 * abs(float(g["loss64"])) + 1.5 * ref_err
    gen, var = model.generator, model.variational
    lq, lpz, lpx = var.nodes["z"].log_prob(), gen.nodes["z"].log_prob(), gen.nodes["x"].log_prob()
    if tag == "small":
        _check_z(var.nodes["z"].dist.sample_cache, g["z"], dev)
    np.testing.assert_allclose((lpz + lpx - lq).detach().cpu().numpy(), g["log_w"], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(lq.detach().cpu().numpy(), g["logqz"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(lpz.detach().cpu().numpy(), g["logpz"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(lpx.detach().cpu().numpy(), g["logpx"], rtol=2e-5, atol=2e-4)


def _seed_philox(dev, seed):
    if dev.type == "cpu":
        host_backend.manual_seed(seed)
    else:
        torch.cuda.manual_seed(seed)            # call id = generator offset // 4 = 0 for the first draw


@pytest.mark.parametrize("tag,B", [("small", 8), ("c2", 512)])
@pytest.mark.parametrize("paired", [True, False])
def test_vae_on_the_in_kernel_philox_stream(dev, tag, B, paired):
    """As test_iwae_on_the_in_kernel_philox_stream for the VAE (ELBO.sgvb, one particle): nothing injected."""
    g = load_golden("g_vae_%s_philox" % tag)
    model = vae_mnist.build(batch_size=B, device=dev)
    H.load_params_into(model, 1000 + B)
    x, _, _ = H.vae_data(B)
    _seed_philox(dev, int(g["philox_seed"]))
    with zs.pair_draws(paired):
        loss = model({"x": T(x, dev)})
    assert rel(loss, g["loss"]) < 2e-5
    gen, var = model.generator, model.variational
    assert rel(gen.nodes["z"].log_prob(), g["logpz"]) < 2e-5
    assert rel(gen.nodes["x"].log_prob(), g["logpx"]) < 2e-5
    assert rel(var.nodes["z"].log_prob(), g["logqz"]) < 2e-5
    model.zero_grad()
    loss.backward()
    _check_grads(g, model, full=(tag == "small"))
    if tag == "small":
        _check_z(var.nodes["z"].dist.sample_cache, g["z"], dev)


@pytest.mark.parametrize("tag,B,K", [("small", 16, 4), ("c5", 512, 10)])
@pytest.mark.parametrize("materialize,paired", [(False, True), (True, True), (True, False)])
def test_bnn_on_the_in_kernel_philox_stream(dev, tag, B, K, materialize, paired):
    """The BNN step (two latents, each drawn twice) with nothing injected: the four draws carry the Philox call ids 0, 1, 2, 3
    in the order the reference makes them (w0, w1, w0 again, w1 again), in every launch mode of the package (one launch for
    all of a pass's draws, or one per draw)."""
    model = bnn_vi.build(n_particles=K, device=dev, materialize=materialize)
    wm, wl, yl = H.bnn_params(B, K)
    with torch.no_grad():
        for i in range(2):
            model.variational.w_means[i].copy_(wm[i])
            model.variational.w_logstds[i].copy_(wl[i])
        model.generator.y_logstd.copy_(yl)
    x, y, _ = H.bnn_data(B, K)
    g = load_golden("g_bnn_%s_philox" % tag)
    _seed_philox(dev, int(g["philox_seed"]))
    with zs.pair_draws(paired):
        loss = model({"x": T(x, dev), "y": T(y, dev)})
    net, var = model.generator, model.variational
    assert rel(loss, g["loss"]) < 2e-5
    assert rel(net.cache["rmse"], g["rmse"]) < 2e-5
    for name in ("w0", "w1"):
        assert rel(net.nodes[name].log_prob(), g["logp_" + name]) < 2e-5
        assert rel(var.nodes[name].log_prob(), g["logq_" + name]) < 2e-5
    assert rel(net.nodes["y"].log_prob(), g["logp_y"]) < 2e-5
    model.zero_grad()
    loss.backward()
    for i in range(2):
        np.testing.assert_allclose(var.w_means[i].grad.cpu().numpy(), g["g_w_mean_%d" % i], rtol=1e-3, atol=5e-5)
        np.testing.assert_allclose(var.w_logstds[i].grad.cpu().numpy(), g["g_w_logstd_%d" % i], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(net.y_logstd.grad.cpu().numpy(), g["g_y_logstd"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("tag,B,K", [("small", 16, 4), ("c5", 512, 10), ("c5g", 4096, 10)])      # c5g: config 5's GLOBAL batch
@pytest.mark.parametrize("materialize", [False, True])
def test_bnn(dev, tag, B, K, materialize):
    g = load_golden("g_bnn_" + tag)
    model = bnn_vi.build(n_particles=K, device=dev, materialize=materialize)
    wm, wl, yl = H.bnn_params(B, K)
    with torch.no_grad():
        for i in range(2):
            model.variational.w_means[i].copy_(wm[i])
            model.variational.w_logstds[i].copy_(wl[i])
        model.generator.y_logstd.copy_(yl)
    x, y, eps = H.bnn_data(B, K)
    with zs.inject_epsilon(eps):                 # draw order w0#1, w1#1, w0#2, w1#2
        loss = model({"x": T(x, dev), "y": T(y, dev)})
    assert rel(loss, g["loss"]) < 2e-5
    net, var = model.generator, model.variational
    assert rel(net.cache["rmse"], g["rmse"]) < 2e-5
    for name in ("w0", "w1"):
        assert rel(net.nodes[name].log_prob(), g["logp_" + name]) < 2e-5
        assert rel(var.nodes[name].log_prob(), g["logq_" + name]) < 2e-5
    assert rel(net.nodes["y"].log_prob(), g["logp_y"]) < 2e-5
    model.zero_grad()
    loss.backward()
    for i in range(2):
        np.testing.assert_allclose(var.w_means[i].grad.cpu().numpy(), g["g_w_mean_%d" % i], rtol=1e-3, atol=5e-5)
        np.testing.assert_allclose(var.w_logstds[i].grad.cpu().numpy(), g["g_w_logstd_%d" % i], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(net.y_logstd.grad.cpu().numpy(), g["g_y_logstd"], rtol=1e-3, atol=1e-4)


def test_training_reduces_loss(dev):
    """A few Adam steps on the Philox path: the surrogate goes down and stays finite."""
    torch.manual_seed(0)
    from zhusuan import _rng
    host_backend.manual_seed(0)
    model = iwae.build(n_samples=8, estimator="vimco", hidden=64, device=dev)
    opt = torch.optim.Adam(model.parameters(), 1e-3)
    x = (torch.rand(32, 784, generator=torch.Generator().manual_seed(1)) < 0.5).float().to(dev)
    bounds = []
    for _ in range(30):
        loss = model({"x": x})
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert torch.isfinite(loss)
        bounds.append(float(model.last_iw_bound.mean()))
    assert np.mean(bounds[-5:]) > np.mean(bounds[:5]) + 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("mod,args", [("examples.vae_mnist", ["--batch", "32", "--steps", "60"]),
                                      ("examples.iwae", ["--batch", "16", "--particles", "8", "--steps", "60"]),
                                      ("examples.iwae", ["--batch", "16", "--particles", "8", "--steps", "60", "--estimator", "sgvb",
                                                         "--fused-logits"]),
                                      ("examples.bnn_vi", ["--steps", "60"]),
                                      ("examples.vae_mnist", ["--batch", "32", "--steps", "60", "--flat-adam"]),
                                      ("examples.bnn_vi", ["--steps", "60", "--flat-adam"]),
                                      ("examples.vae_mnist", ["--batch", "32", "--steps", "60", "--flat-adam", "--dense", "fused"]),
                                      ("examples.iwae", ["--batch", "16", "--particles", "8", "--steps", "60", "--dense", "fused"]),
                                      ("examples.bnn_vi", ["--steps", "60", "--layer", "per_layer"])])
def test_example_scripts_run(mod, args):
    """The counterparts of the reference's example scripts run as programs on the GPU (synthetic data)."""
    import subprocess
    import sys
    from conftest import ROOT, PKG_ROOT
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG_ROOT, ROOT]))
    r = subprocess.run([sys.executable, "-m", mod] + args, cwd=PKG_ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "ELBO-evals/s" in r.stdout and "nan" not in r.stdout.lower()


# ------------------------------------------------------------------ seed-compatible mode
def test_seeded_runs_reproduce_the_reference(dev):
    """``torch.manual_seed(s)`` + ``zhusuan.reference_rng()``: every draw is made on the host from torch's CPU generator
    with the reference's own call, in the reference's order, so a seeded run reproduces the seeded run of the reference
    (golden: tests/golden/gen_golden.py gen_seeded, no draw injected there)."""
    g = load_golden("g_seeded")
    seed = int(g["seed"])
    B = 8
    model = vae_mnist.build(batch_size=B, device=dev)
    H.load_params_into(model, 1000 + B)
    x, _, _ = H.vae_data(B)
    torch.manual_seed(seed)
    with zs.reference_rng():
        loss = model({"x": T(x, dev)})
    assert rel(loss, g["vae_loss"]) < 2e-5
    _check_z(model.variational.nodes["z"].dist.sample_cache, g["vae_z"], dev)
    model.zero_grad()
    loss.backward()
    norms = np.array([float(p.grad.double().norm()) for p in model.parameters()])
    np.testing.assert_allclose(norms, g["vae_grad_norms"], rtol=1e-3, atol=1e-6)
    B, K = 8, 5
    for est in ("sgvb", "vimco"):
        model = iwae.build(n_samples=K, estimator=est, hidden=32, device=dev)
        H.load_params_into(model, 2000 + B + K)
        x, _, _ = H.iwae_data(B, K)
        torch.manual_seed(seed)
        with zs.reference_rng():
            loss = model({"x": T(x, dev)})
        assert rel(loss, g["iwae_%s_loss" % est]) < 5e-5, est
        _check_z(model.variational.nodes["z"].dist.sample_cache, g["iwae_%s_z" % est], dev)
        model.zero_grad()
        loss.backward()
        norms = np.array([float(p.grad.double().norm()) for p in model.parameters()])
        np.testing.assert_allclose(norms, g["iwae_%s_grad_norms" % est], rtol=1e-3, atol=1e-6)
    B, K = 16, 4
    model = bnn_vi.build(n_particles=K, device=dev)
    wm, wl, yl = H.bnn_params(B, K)
    with torch.no_grad():
        for i in range(2):
            model.variational.w_means[i].copy_(wm[i])
            model.variational.w_logstds[i].copy_(wl[i])
        model.generator.y_logstd.copy_(yl)
    x, y, _ = H.bnn_data(B, K)
    torch.manual_seed(seed)
    with zs.reference_rng():
        loss = model({"x": T(x, dev), "y": T(y, dev)})
    assert rel(loss, g["bnn_loss"]) < 2e-5
    model.zero_grad()
    loss.backward()
    np.testing.assert_allclose(model.variational.w_means[0].grad.cpu().numpy(), g["bnn_g_w_mean_0"], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(model.variational.w_logstds[1].grad.cpu().numpy(), g["bnn_g_w_logstd_1"], rtol=1e-3, atol=5e-5)
