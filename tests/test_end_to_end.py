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


def _check_grads(g, model, rtol_norm=5e-4):
    names = [str(n) for n in g["grad_names"]]
    named = list(model.named_parameters())
    assert names == [n for n, _ in named]
    norms = np.array([float(p.grad.double().norm()) for _, p in named])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=rtol_norm, atol=1e-6)
    for n, p in named:
        np.testing.assert_allclose(p.grad.reshape(-1)[:8].cpu().numpy(), g["ghead_" + n], rtol=5e-3, atol=5e-5)


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
    _check_grads(g, model)
    if tag == "small":
        _check_z(var.nodes["z"].dist.sample_cache, g["z"], dev)
        np.testing.assert_allclose(gen.cache["x_mean"].detach().cpu().numpy(), g["x_mean"], rtol=1e-4, atol=1e-6)
    # generation path (vae_mnist.py:129-136): fresh prior sample, Bernoulli draw
    gen({})
    assert gen.cache["x_mean"].shape == (B, 784)
    assert set(np.unique(gen.nodes["x"].tensor.cpu().numpy())) <= {0.0, 1.0}


@pytest.mark.parametrize("est", ["sgvb", "vimco"])
@pytest.mark.parametrize("tag,B,K,hidden", [("small", 8, 5, 32), ("c3", 256, 50, 500)])
@pytest.mark.parametrize("fused_logits", [False, True])
def test_iwae(dev, est, tag, B, K, hidden, fused_logits):
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
    _check_grads(g, model, rtol_norm=2e-3)
    gen, var = model.generator, model.variational
    lq = var.nodes["z"].log_prob()
    assert tuple(lq.shape) == (K, B) and lq.stride() == (1, K)      # K-fastest rows, reference shape
    log_w = (gen.nodes["z"].log_prob() + gen.nodes["x"].log_prob() - lq).detach().cpu().numpy()
    if tag == "small":
        _check_z(var.nodes["z"].dist.sample_cache, g["z"], dev)
        np.testing.assert_allclose(log_w, g["log_w"], rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(lq.detach().cpu().numpy(), g["logqz"], rtol=2e-5, atol=2e-5)
    else:
        np.testing.assert_allclose(log_w[:, 0], g["log_w_col0"], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("tag,B,K", [("small", 16, 4), ("c5", 512, 10)])
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
    _rng.manual_seed_host(0)
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
                                      ("examples.bnn_vi", ["--steps", "60"])])
def test_example_scripts_run(mod, args):
    """The counterparts of the reference's example scripts run as programs on the GPU (synthetic data)."""
    import subprocess
    import sys
    from conftest import ROOT, PKG_ROOT
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG_ROOT, ROOT]))
    r = subprocess.run([sys.executable, "-m", mod] + args, cwd=PKG_ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "ELBO-evals/s" in r.stdout and "nan" not in r.stdout.lower()
