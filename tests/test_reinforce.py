"""ELBO.reinforce: the fused epilogue kernel R1 against the reference.

Goldens: tests/golden/g_elbo_reinforce.npz -- 10 configurations x 3 consecutive calls of the REAL reference's
``ELBO.reinforce`` (the moving mean and the step counter are state), with gradients.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden, host_kernel_library
from oracle import zs_oracle as O
from zhusuan import _hip
from zhusuan.variational.elbo import ELBO


def T(a, dev="cpu", rg=False):
    x = torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    return x.requires_grad_(rg)


def close(a, b, rtol=2e-6, atol=2e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def configs():
    g = load_golden("g_elbo_reinforce")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        steps = []
        for s in range(3):
            q = p + "s%d_" % s
            steps.append({k[len(q):]: g[k] for k in g.files if k.startswith(q)})
        yield c, bool(g[p + "vr"]), bool(g[p + "rm"]), int(g[p + "bkind"]), steps


def _check_steps(run, dev, rtol, atol):
    """run(step dict, logp, logq, baseline) -> loss (and maybe elbo mean); state lives in the closure's object."""
    n = 0
    for c, vr, rm, bkind, steps in configs():
        state = run("new", None, None, None, None, None)
        for st in steps:
            lp, lq = T(st["logp"], dev, True), T(st["logq"], dev, True)
            base = T(st["baseline"], dev, True) if "baseline" in st else None
            res = run(state, lp, lq, base, vr, rm)
            loss = res[0] if isinstance(res, tuple) else res
            assert tuple(loss.shape) == st["loss"].shape, (c, tuple(loss.shape), st["loss"].shape)
            close(loss, st["loss"], rtol, atol * max(1.0, float(np.abs(st["loss"]).max())))
            if isinstance(res, tuple):
                close(res[1], st["elbo_mean"], 2e-6, 1e-4)
            else:
                assert "elbo_mean" not in st
            inputs = [lp, lq] + ([base] if (base is not None and vr) else [])
            grads = torch.autograd.grad((loss * T(st["w"], dev)).sum(), inputs, allow_unused=True)
            gp = grads[0] if grads[0] is not None else torch.zeros_like(lp)
            close(gp, st["glogp"], 1e-5, 1e-6)
            close(grads[1], st["glogq"], 2e-5, 2e-5 * max(1.0, float(np.abs(st["glogq"]).max())))
            if base is not None and vr:
                close(grads[2], st["gbaseline"], 2e-5, 2e-5 * max(1.0, float(np.abs(st["gbaseline"]).max())))
            mm, ls = state_values(state)
            close(mm, st["moving_mean"], 2e-6, 2e-5)
            assert int(ls) == int(np.asarray(st["local_step"]).reshape(-1)[0])
            n += 1
    assert n == 30


def state_values(state):
    if isinstance(state, ELBO):
        return state.moving_mean.detach().cpu().numpy(), int(state.local_step)
    return state[0].numpy(), int(state[1])


def test_oracle_reinforce_golden():
    def run(state, lp, lq, base, vr, rm):
        if state == "new":
            return (torch.zeros(1), torch.zeros(1, dtype=torch.int32))
        return O.elbo_reinforce(lp, lq, state[0], state[1], reduce_mean=rm, baseline=base, variance_reduction=vr, decay=0.8)
    _check_steps(run, "cpu", 2e-6, 2e-6)


def test_product_reinforce_golden(dev):
    def run(state, lp, lq, base, vr, rm):
        if state == "new":
            return ELBO(None, None, estimator="reinforce").to(dev)
        return state.reinforce(lp, lq, reduce_mean=rm, baseline=base, variance_reduction=vr, decay=0.8)
    _check_steps(run, dev, 1e-5, 2e-6)


def test_reinforce_shape_errors_match_the_reference(dev):
    e = ELBO(None, None, estimator="reinforce").to(dev)
    z0 = torch.zeros((), device=dev)
    with pytest.raises(RuntimeError, match=r"output with shape \[\] doesn't match the broadcast shape \[1\]"):
        e.reinforce(z0, z0)                                           # elbo.py:225 on a 0-d log-joint
    with pytest.raises(RuntimeError, match=r"output with shape \[1\] doesn't match the broadcast shape \[5\]"):
        e.reinforce(torch.zeros(5, device=dev), torch.zeros(5, device=dev), reduce_mean=False)   # elbo.py:221
    assert float(e.reinforce(z0 + 1, z0 + 2, variance_reduction=False)) == -(1.0 + (1.0 - 2.0) * 2.0)
    assert int(e.local_step) == 0


def _raw(klib, dev, logp, logq, base, vr, do_mean, mm, step, dtype=torch.float32, Pb=None, workspace=True):
    sfx = "_f32" if dtype == torch.float32 else "_f64"
    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)
    lp, lq, b = t(logp), t(logq), t(base)
    n = lq.numel()
    mmt = torch.tensor([mm], dtype=torch.float32, device=dev)
    stt = torch.tensor([step], dtype=torch.int32, device=dev)
    sig = torch.full((n,), float("nan"), dtype=dtype, device=dev)
    res = torch.full((n,), float("nan"), dtype=dtype, device=dev)
    cost = torch.full((1 if do_mean else n,), float("nan"), dtype=dtype, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if torch.device(dev).type == "cuda" else None
    # workspace + ticket: long vectors then take the many-workgroup path (twice on one workspace: the ticket must come back
    # at zero and the second call must not see the first one's partial sums)
    ws = torch.zeros(_hip.LJ_WORKSPACE, dtype=torch.float64, device=dev) if workspace else None
    tk = torch.zeros(1, dtype=torch.int32, device=dev) if workspace else None
    for rep in range(2 if workspace else 1):
        mmt.fill_(mm)
        stt.fill_(step)
        first = rep == 0 and workspace
        klib.call("zs_reinforce" + sfx, _hip.ptr(lp * 0.5 if first else lp), _hip.ptr(lq), _hip.ptr(b),
                  Pb if Pb is not None else (1 if (b is None or b.numel() == 1) else n), n,
                  int(vr), int(do_mean), 0.8, _hip.ptr(mmt), _hip.ptr(stt), _hip.ptr(sig), _hip.ptr(cost),
                  _hip.ptr(res) if b is not None else None, _hip.ptr(ws), _hip.LJ_WORKSPACE if workspace else 0, _hip.ptr(tk), st)
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()
    assert tk is None or int(tk.item()) == 0
    return dict(cost=cost.cpu().numpy(), signal=sig.cpu().numpy(), resid=res.cpu().numpy() if b is not None else None,
                mm=float(mmt), step=int(stt))


def test_c_oracle_rejects_bad_reinforce_arguments():
    k = host_kernel_library()
    one = np.ones(4, np.float32)
    with pytest.raises(RuntimeError, match="code -1"):
        _raw(k, "cpu", one, one, None, True, False, 0.0, 0)          # vector without mean and variance reduction
    with pytest.raises(RuntimeError, match="code -1"):
        _raw(k, "cpu", one, one, np.ones(4, np.float32), True, True, 0.0, 0, Pb=2)   # baseline period neither 1 nor n


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 64, 65, 300, 1024, 5000, 16384, 16385, 70001, 1000003])
@pytest.mark.parametrize("bkind", [0, 1, 2])
@pytest.mark.parametrize("vr", [True, False])
def test_hip_reinforce_vs_c_oracle(n, bkind, vr):
    rng = np.random.RandomState(n + bkind)
    hipk, orc = _hip.KernelLibrary(_hip.LIB_PATH), host_kernel_library()
    logp = (-90 + 3 * rng.standard_normal(n)).astype(np.float32)
    logq = (-40 + rng.standard_normal(n)).astype(np.float32)
    base = None if bkind == 0 else ((-50 + rng.standard_normal(n if bkind == 1 else 1)).astype(np.float32))
    for do_mean in ([True, False] if (n == 1 or not vr) else [True]):
        a = _raw(hipk, "cuda:0", logp, logq, base, vr, do_mean, 0.3, 2)
        b = _raw(orc, "cpu", logp, logq, base, vr, do_mean, 0.3, 2)
        np.testing.assert_allclose(a["cost"], b["cost"], rtol=2e-6, atol=1e-3 if not do_mean else 2e-4)
        np.testing.assert_allclose(a["signal"], b["signal"], rtol=1e-6, atol=2e-5)
        if base is not None and vr:
            np.testing.assert_allclose(a["resid"], b["resid"], rtol=1e-6, atol=1e-5)
        assert abs(a["mm"] - b["mm"]) <= 2e-6 * max(1.0, abs(b["mm"])) and a["step"] == b["step"] == (3 if vr else 2)
        if n > 16384:       # without a workspace one workgroup walks the vector: same numbers
            c = _raw(hipk, "cuda:0", logp, logq, base, vr, do_mean, 0.3, 2, workspace=False)
            np.testing.assert_allclose(c["cost"], a["cost"], rtol=2e-6, atol=2e-4)
            np.testing.assert_allclose(c["signal"], a["signal"], rtol=1e-6, atol=2e-5)
    for dt in (torch.float64,):
        a = _raw(hipk, "cuda:0", logp, logq, base, vr, True, 0.3, 2, dt)
        b = _raw(orc, "cpu", logp, logq, base, vr, True, 0.3, 2, dt)
        # the moving mean is a float32 buffer in the reference (elbo.py:45), so its last ulp (device vs host powf)
        # bounds the agreement of the float64 variant whenever variance reduction is on
        tol = 1e-6 if vr else 1e-12
        np.testing.assert_allclose(a["cost"], b["cost"], rtol=tol, atol=1e-9 if not vr else 1e-3)
        np.testing.assert_allclose(a["signal"], b["signal"], rtol=tol, atol=1e-9 if not vr else 1e-4)


@pytest.mark.gpu
def test_reinforce_is_graph_capturable():
    dev = torch.device("cuda:0")
    e = ELBO(None, None, estimator="reinforce").to(dev)
    lp = torch.randn(64, device=dev) - 90
    lq = (torch.randn(64, device=dev) - 40).requires_grad_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        e.reinforce(lp, lq).backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    lq.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cost = e.reinforce(lp, lq)
        cost.backward()
    step0 = int(e.local_step)
    vals = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals.append((float(cost), float(e.moving_mean)))
    assert int(e.local_step) == step0 + 3                   # device-side state advances on every replay
    assert len({v[1] for v in vals}) == 3
