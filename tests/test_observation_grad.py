"""Gradients w.r.t. a Bernoulli OBSERVATION (missing 2): the reference's ``Bernoulli._log_prob``
(zhusuan/distributions/bernoulli.py:84-95) is differentiable in ``sample`` -- d/dx = log(p + 1e-8) - log(1 - p + 1e-8) -- and
``given`` keeps its graph through ``Distribution.log_prob`` (base.py:161-178), so a model whose observed value comes out of a
differentiable net trains in the reference.  Here the three paths that used to raise -- the per-node kernel (K3), the one-launch
generator side of the importance-weighted objective (IW1) and the one-launch scalar log-joint (LJ1) -- against
``g_observation_grad.npz``, captured from the real reference (tests/golden/gen_golden.py::gen_observation_grad).
"host" back-end on CPU (the C oracle behind the same ABI) and, marked gpu, the HIP library."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden, host_kernel_library
import helpers as H
import zhusuan as zs
from zhusuan import _hip
from examples import vae_mnist, iwae


def T(a, dev, grad=False):
    x = torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    return x.requires_grad_(grad)


def _close(got, ref, rtol, atol_of_max):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_of_max * max(float(np.abs(ref).max()), 1e-30))


def test_distribution_level(dev):
    g = load_golden("g_observation_grad")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        par, x = T(g[p + "param"], dev, True), T(g[p + "x"], dev, True)
        d = zs.distributions.Bernoulli(logits=par, group_ndims=int(g[p + "g"])) if int(g[p + "from_logits"]) else \
            zs.distributions.Bernoulli(probs=par, group_ndims=int(g[p + "g"]))
        lp = d.log_prob(x)
        _close(lp, g[p + "lp"], 2e-5, 2e-6)
        gpar, gx = torch.autograd.grad((lp * T(g[p + "w"], dev)).sum(), [par, x])
        assert gx.shape == x.shape
        _close(gpar, g[p + "gparam"], 1e-4, 1e-6)
        _close(gx, g[p + "gx"], 1e-4, 2e-6)         # sums over the K rows that read an observation: ascending, as the oracle
        # only the observation's gradient wanted: the parameter's launch is skipped, same numbers
        (gx2,) = torch.autograd.grad((d.log_prob(x) * T(g[p + "w"], dev)).sum(), [x])
        assert torch.equal(gx2, gx)


def _spy(dev):
    klib = host_kernel_library() if dev.type == "cpu" else _hip.lib()
    calls, real = [], klib.call
    klib.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    return klib, calls, real


@pytest.mark.parametrize("est", ["sgvb", "vimco"])
@pytest.mark.parametrize("fused_logits", [False, True])
def test_iwae_with_a_differentiable_observation(dev, est, fused_logits):
    """The IWAE example with x a leaf that requires a gradient: d loss / d x through the likelihood (IW1's backward: the
    observation's own launch, row gradients coef * g formed in the kernel) and through the encoder."""
    g = load_golden("g_observation_grad")
    pre = "iwae_%s_" % est
    B, K, hidden = [int(v) for v in g[pre + "shape"]]
    model = iwae.build(n_samples=K, estimator=est, hidden=hidden, device=dev, fused_logits=fused_logits)
    H.load_params_into(model, 2000 + B + K)
    x = T(g[pre + "x"], dev, True)
    klib, calls, real = _spy(dev)
    try:
        with zs.inject_epsilon([g[pre + "e1"], g[pre + "e2"]]):
            loss = model({"x": x})
        model.zero_grad()
        loss.backward()
    finally:
        klib.call = real
    assert abs(float(loss.detach()) - float(g[pre + "loss"])) < 5e-5 * abs(float(g[pre + "loss"]))
    assert "zs_bernoulli_iw_objective_f32" in calls and calls.count("zs_bernoulli_logprob_bwd_x_f32") == 1      # the fused path
    _close(x.grad, g[pre + "gx"], 2e-3, 3e-4)
    names = [str(n) for n in g[pre + "grad_names"]]
    assert names == [n for n, _ in model.named_parameters()]
    norms = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    np.testing.assert_allclose(norms, g[pre + "grad_norms"], rtol=2e-3, atol=1e-6)


def test_vae_with_a_differentiable_observation(dev):
    """The VAE's scalar ELBO (every node's log-probability in one launch each way, LJ1): the Bernoulli term's observation
    receives its gradient inside that launch."""
    g = load_golden("g_observation_grad")
    B = 8
    model = vae_mnist.build(batch_size=B, device=dev)
    H.load_params_into(model, 1000 + B)
    x = T(g["vae_x"], dev, True)
    klib, calls, real = _spy(dev)
    try:
        with zs.inject_epsilon([g["vae_e1"], g["vae_e2"]]):
            loss = model({"x": x})
        model.zero_grad()
        loss.backward()
    finally:
        klib.call = real
    assert abs(float(loss.detach()) - float(g["vae_loss"])) < 2e-5 * abs(float(g["vae_loss"]))
    assert "zs_logjoint_scalar_bwd_f32" in calls and "zs_bernoulli_logprob_bwd_x_f32" not in calls         # inside LJ1's launch
    _close(x.grad, g["vae_gx"], 1e-3, 1e-4)
    norms = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    np.testing.assert_allclose(norms, g["vae_grad_norms"], rtol=1e-3, atol=1e-6)


def _lib_and_ptr(dev):
    if dev.type == "cpu":
        return host_kernel_library(), None
    return _hip.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("K,R,D,Px,strides", [(5, 7, 12, 7 * 12, "kfast"), (3, 4, 8, 3 * 4 * 8, "rows"), (4, 6, 10, 10, "kfast"),
                                                (3, 5, 8, 20, "kfast"), (2, 3, 7, 1, "rows"), (50, 16, 784, 16 * 784, "kfast")])
@pytest.mark.parametrize("from_logits", [0, 1])
def test_cabi_observation_gradient_against_float64(dev, K, R, D, Px, strides, from_logits):
    """zs_bernoulli_logprob_bwd_x through the raw C ABI: periods that are whole rows, a single row, a fraction of a row (the
    division path), a scalar; K-fastest and row-major row gradients; with and without the device-resident scale -- against
    the formula in float64."""
    lib, st = _lib_and_ptr(dev)
    rng = np.random.RandomState(K * 100 + R * 10 + D + Px)
    n = K * R * D
    par = (2.0 * rng.standard_normal(n) if from_logits else rng.uniform(0.01, 0.99, n)).astype(np.float32)
    glp = rng.standard_normal((K, R)).astype(np.float32)
    scale = rng.standard_normal(R).astype(np.float32)
    p64 = 1.0 / (1.0 + np.exp(-par.astype(np.float64))) if from_logits else par.astype(np.float64)
    if from_logits:
        p64 = (1.0 / (1.0 + np.exp(-par))).astype(np.float32).astype(np.float64)          # (the kernels form p in fp32)
    term = np.log(p64 + 1e-8) - np.log((1.0 - p64) + 1e-8)
    for use_scale in (False, True):
        rowg = glp.astype(np.float64) * (scale.astype(np.float64)[None, :] if use_scale else 1.0)
        ref = (np.repeat(rowg.reshape(-1), D) * term).reshape(n // Px, Px).sum(0)
        gl = T(glp if strides == "rows" else glp.T.copy(), dev)
        sk, sr = (R, 1) if strides == "rows" else (1, K)
        pt, sc, gx = T(par, dev), T(scale, dev), torch.full((Px,), float("nan"), device=dev)
        lib.call("zs_bernoulli_logprob_bwd_x_f32", _hip.ptr(pt), from_logits, Px, _hip.ptr(gl), sk, sr,
                 _hip.ptr(sc) if use_scale else None, 1 if use_scale else 0, _hip.ptr(gx), K, R, D, st)
        _close(gx, ref, 2e-4, 2e-5)
