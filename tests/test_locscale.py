"""Logistic and Uniform: oracle pins, C-ABI parity and the product classes.

  * oracle/zs_oracle.py and oracle/zs_oracle_c.c against the goldens generated from the real reference
    (tests/golden/g_logistic_*.npz, g_uniform_*.npz)  -- CPU;
  * libzs_hip.so against the C oracle, entry point by entry point                              -- gpu;
  * zhusuan.distributions.Logistic / Uniform on the "host" back-end (CPU) and on the HIP library (gpu): the
    reference's own unit tests restated (test/distributions/test_logistic.py, test_uniform.py) + goldens.
"""
import numpy as np
import pytest
import torch
from scipy import stats

from conftest import load_golden, host_kernel_library
import host_backend
from oracle import zs_oracle as O
from test_cabi import Raw
from zhusuan import _hip
import zhusuan as zs
from zhusuan.distributions import Logistic, Uniform
from zhusuan.framework.bn import BayesianNet


def T(a, dev="cpu", rg=False, dtype=np.float32):
    x = torch.tensor(np.asarray(a, dtype=dtype), device=dev)
    return x.requires_grad_(rg)


def close(a, b, rtol=1e-5, atol=2e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def cases(name):
    g = load_golden(name)
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        yield c, {k[len(p):]: g[k] for k in g.files if k.startswith(p)}


# ------------------------------------------------------------------ python oracle vs goldens
def test_oracle_logistic_sample():
    n = 0
    for c, g in cases("g_logistic_sample"):
        K = None if int(g["K"]) < 0 else int(g["K"])
        loc, sc = T(g["loc"], rg=True), T(g["scale"], rg=True)
        z = O.logistic_sample(loc, sc, T(g["u"]), K)
        lp = O.logistic_log_prob(loc, sc, z, int(g["g"]))
        assert np.array_equal(z.detach().numpy(), g["z"]), "case %d" % c
        close(lp, g["lp"], 2e-6, 2e-6)
        gl, gs = torch.autograd.grad((lp * T(g["w"])).sum() + (z * T(g["wz"])).sum(), [loc, sc])
        close(gl, g["gloc"], 2e-5, 2e-5)
        close(gs, g["gscale"], 2e-5, 2e-5)
        n += 1
    assert n == 18


def test_oracle_logistic_logprob():
    for c, g in cases("g_logistic_logprob"):
        loc, sc, x = T(g["loc"], rg=True), T(g["scale"], rg=True), T(g["x"], rg=True)
        lp = O.logistic_log_prob(loc, sc, x, int(g["g"]))
        close(lp, g["lp"], 2e-6, 2e-6)
        gl, gs, gx = torch.autograd.grad((lp * T(g["w"])).sum(), [loc, sc, x])
        close(gl, g["gloc"], 1e-5, 1e-5)
        close(gs, g["gscale"], 1e-5, 1e-5)
        close(gx, g["gx"], 1e-5, 1e-5)
    # reference test_logistic.py:69-74 (scipy logpdf, rtol 1e-3)
    kat = load_golden("g_logistic_logprob")["kat_lp"]
    close(kat, stats.logistic.logpdf([3.], [2.], [1.]), 1e-6, 0)
    close(O.logistic_log_prob(T([2.]), T([1.]), T([3.])), kat, 1e-6, 0)


def test_oracle_uniform():
    n = 0
    for c, g in cases("g_uniform_sample"):
        K = None if int(g["K"]) < 0 else int(g["K"])
        low, high = T(g["low"], rg=True), T(g["high"], rg=True)
        assert tuple(g["draw_shape"]) == g["u"].shape
        z, cache = O.uniform_sample(low, high, T(g["u"]), K, bool(g["reparam"]))
        assert np.array_equal(z.detach().numpy(), g["z"]) and np.array_equal(cache.detach().numpy(), g["cache"])
        glo, ghi = torch.autograd.grad((z * T(g["wz"])).sum(), [low, high])
        close(glo, g["glow"], 2e-6, 2e-6)
        close(ghi, g["ghigh"], 2e-6, 2e-6)
        n += 1
    assert n == 12
    g = load_golden("g_uniform_sample")
    assert tuple(g["bc_draw_shape"]) == (2, 1, 3)           # the draw has LOW's shape
    z, _ = O.uniform_sample(T(g["bc_low"]), T(g["bc_high"]), T(g["bc_u"]), 2)
    assert np.array_equal(z.numpy(), g["bc_z"])
    for c, g in cases("g_uniform_logprob"):
        low, high = T(g["low"], rg=True), T(g["high"], rg=True)
        lp = O.uniform_log_prob(low, high, T(g["x"]), int(g["g"]))
        close(lp, g["lp"], 1e-6, 1e-6)
        glo, ghi = torch.autograd.grad((lp * T(g["w"])).sum(), [low, high])
        close(glo, g["glow"], 1e-5, 1e-5)
        close(ghi, g["ghigh"], 1e-5, 1e-5)
    kat = load_golden("g_uniform_logprob")["kat_lp"]
    assert kat.dtype == np.float64 and float(kat[0]) == 0.0     # test_uniform.py:80: logpdf(4.5; 4, 5) = 0


# ------------------------------------------------------------------ raw C-ABI helpers
class Raw2(Raw):
    def logistic_sample(self, loc, scale, u, K, D, seed=0, off=0, kfast=False, want_lp=True, rs=None, used=None):
        M = loc.size
        R = M // D
        z = self.empty(K, M)
        lp = self.empty(R, K) if kfast else self.empty(K, R)
        sk, sr = (1, K) if kfast else (R, 1)
        self.call("zs_logistic_sample_logprob_f32", self.t(loc), self.t(scale), self.t(u), seed, off, rs, z,
                  lp if want_lp else None, K, M, D, sk, sr, used)
        lpn = lp.cpu().numpy()
        return dict(z=z.cpu().numpy(), lp=lpn.T if kfast else lpn)

    def logistic_sample_bwd(self, scale, u, gz, glp, K, D, seed=0, off=0, rs=None):
        M = scale.size
        gl, gs = self.empty(M), self.empty(M)
        self.call("zs_logistic_sample_logprob_bwd_f32", self.t(scale), self.t(u), seed, off, rs, self.t(gz), self.t(glp),
                  M // D, 1, gl, gs, K, M, D)
        return dict(gloc=gl.cpu().numpy(), gscale=gs.cpu().numpy())

    def _rows(self, name, a, b, c, K, R, D, kfast):
        lp = self.empty(R, K) if kfast else self.empty(K, R)
        sk, sr = (1, K) if kfast else (R, 1)
        self.call(name, self.t(a), a.size, self.t(b), b.size, self.t(c), c.size, lp, K, R, D, sk, sr)
        lpn = lp.cpu().numpy()
        return dict(lp=lpn.T if kfast else lpn)

    def logistic_lp(self, x, loc, scale, K, R, D, kfast=False):
        return self._rows("zs_logistic_logprob_f32", x, loc, scale, K, R, D, kfast)

    def uniform_lp(self, x, low, high, K, R, D, kfast=False):
        return self._rows("zs_uniform_logprob_f32", x, low, high, K, R, D, kfast)

    def logistic_lp_bwd(self, x, loc, scale, glp, K, R, D, want=(True, True, True)):
        N = K * R * D
        outs = [self.empty(N) if w else None for w in want]
        self.call("zs_logistic_logprob_bwd_f32", self.t(x), x.size, self.t(loc), loc.size, self.t(scale), scale.size,
                  self.t(glp), R, 1, outs[0], outs[1], outs[2], K, R, D)
        return {n: o.cpu().numpy() for n, o in zip(("gx", "gloc", "gscale"), outs) if o is not None}

    def logistic_lp_bwd_ksum(self, x, loc, scale, glp, K, R, D, want_gx=True, kfast=False):
        """glp [K, R] (or its K-fastest transpose with kfast): strides as the caller's objective leaves them."""
        gx = self.empty(K * R * D) if want_gx else None
        gloc, gscale = self.empty(R * D), self.empty(R * D)
        g = np.ascontiguousarray(glp.reshape(K, R).T) if kfast else glp
        sk, sr = (1, K) if kfast else (R, 1)
        self.call("zs_logistic_logprob_bwd_ksum_f32", self.t(x), self.t(loc), self.t(scale), self.t(g), sk, sr, gx, gloc, gscale,
                  K, R, D)
        out = dict(gloc=gloc.cpu().numpy(), gscale=gscale.cpu().numpy())
        if want_gx:
            out["gx"] = gx.cpu().numpy()
        return out

    def uniform_sample(self, low, high, u, N, reparam, seed=0, off=0, rs=None, want_cache=True):
        out, cache = self.empty(N), self.empty(N)
        self.call("zs_uniform_sample_f32", self.t(low), low.size, self.t(high), high.size, self.t(u), seed, off, rs, out,
                  cache if want_cache else None, N, int(reparam))
        return dict(out=out.cpu().numpy(), cache=cache.cpu().numpy())

    def philox_u(self, n, seed, off, rs=None):
        out = self.empty(n)
        self.call("zs_philox_uniform_f32", out, n, seed, off, rs)
        return out.cpu().numpy()


@pytest.fixture(scope="module")
def orc():
    return Raw2(host_kernel_library(), "cpu")


@pytest.fixture(scope="module")
def hip():
    return Raw2(_hip.KernelLibrary(_hip.LIB_PATH), "cuda:0")


@pytest.fixture(scope="module")
def orc64():
    return Raw2(host_kernel_library(), "cpu", torch.float64)


@pytest.fixture(scope="module")
def hip64():
    return Raw2(_hip.KernelLibrary(_hip.LIB_PATH), "cuda:0", torch.float64)


# ------------------------------------------------------------------ C oracle vs goldens (CPU)
def test_c_oracle_logistic_golden(orc):
    for c, g in cases("g_logistic_sample"):
        K = max(int(g["K"]), 1)
        loc, sc, u = g["loc"], g["scale"], g["u"]
        gnd = int(g["g"])
        D = int(np.prod(loc.shape[loc.ndim - gnd:])) if gnd else 1
        out = orc.logistic_sample(loc.ravel(), sc.ravel(), u.ravel(), K, D)
        # libm logf vs torch's vectorised log: an ulp apart, so z is close rather than bit-identical
        np.testing.assert_allclose(out["z"].reshape(g["z"].shape), g["z"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(out["lp"].reshape(g["lp"].shape), g["lp"], rtol=1e-5, atol=1e-5)
    for c, g in cases("g_logistic_logprob"):
        loc, sc, x = g["loc"], g["scale"], g["x"]
        if not (loc.shape == sc.shape and x.shape[x.ndim - loc.ndim:] == loc.shape):
            continue                                   # middle-axis broadcasts are materialised by the host code
        gnd = int(g["g"])
        D = int(np.prod(x.shape[x.ndim - gnd:])) if gnd else 1
        rows = x.size // D
        out = orc.logistic_lp(x.ravel(), loc.ravel(), sc.ravel(), 1, rows, D)
        np.testing.assert_allclose(out["lp"].reshape(g["lp"].shape), g["lp"], rtol=1e-5, atol=1e-5)
        if gnd == 0:
            b = orc.logistic_lp_bwd(x.ravel(), loc.ravel(), sc.ravel(), g["w"].ravel(), 1, rows, 1)
            np.testing.assert_allclose(b["gx"].reshape(x.shape), g["gx"], rtol=2e-5, atol=2e-6)
            rep = x.size // loc.size
            np.testing.assert_allclose(b["gloc"].reshape(rep, -1).sum(0).reshape(loc.shape), g["gloc"], rtol=2e-5, atol=1e-5)
            np.testing.assert_allclose(b["gscale"].reshape(rep, -1).sum(0).reshape(sc.shape), g["gscale"], rtol=2e-5, atol=1e-5)


def test_c_oracle_uniform_golden(orc):
    for c, g in cases("g_uniform_sample"):
        low, high, u = g["low"], g["high"], g["u"]
        out = orc.uniform_sample(low.ravel(), high.ravel(), u.ravel(), u.size, bool(g["reparam"]))
        assert np.array_equal(out["out"].reshape(g["z"].shape), g["z"])
        assert np.array_equal(out["cache"].reshape(g["cache"].shape), g["cache"])
    for c, g in cases("g_uniform_logprob"):
        low, high, x = g["low"], g["high"], g["x"]
        if not (low.shape == high.shape and x.shape[x.ndim - low.ndim:] == low.shape):
            continue
        gnd = int(g["g"])
        D = int(np.prod(x.shape[x.ndim - gnd:])) if gnd else 1
        out = orc.uniform_lp(x.ravel(), low.ravel(), high.ravel(), 1, x.size // D, D)
        np.testing.assert_allclose(out["lp"].reshape(g["lp"].shape), g["lp"], rtol=1e-6, atol=1e-6)


def test_c_oracle_uniform_outside_support_and_bad_args(orc):
    out = orc.uniform_lp(np.array([0.5, 2.0, -1.0, 1.0], np.float32), np.zeros(1, np.float32), np.ones(1, np.float32), 1, 4, 1)
    assert out["lp"].ravel().tolist() == [0.0, -np.inf, -np.inf, -np.inf]          # high is exclusive
    out = orc.uniform_lp(np.array([0.5, 2.0, 0.1, 0.2], np.float32), np.zeros(1, np.float32), np.ones(1, np.float32), 1, 2, 2)
    assert out["lp"].ravel().tolist() == [-np.inf, 0.0]
    one = np.ones(4, np.float32)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.logistic_sample(one, one, one, 1, 3)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.logistic_lp(np.ones(3, np.float32), one, one, 1, 1, 4)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.uniform_sample(np.ones(3, np.float32), one, None, 4, True)


def test_c_oracle_philox_uniform(orc):
    u = orc.philox_u(1 << 16, 9, 4)
    assert 0.0 < u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1.0 / 12) < 2e-3
    assert not np.array_equal(u, orc.philox_u(1 << 16, 9, 5))
    assert np.array_equal(u[:1000], orc.philox_u(1000, 9, 4))
    # device-resident state: seed / base come from rng_state
    rs = torch.tensor([9, 3], dtype=torch.int64)
    assert np.array_equal(orc.philox_u(1000, 0, 1, rs), u[:1000])
    # the draw of L1 / U1 is this stream
    a = orc.uniform_sample(np.zeros(1, np.float32), np.ones(1, np.float32), None, 1000, True, seed=9, off=4)
    assert np.array_equal(a["cache"], u[:1000])


# ------------------------------------------------------------------ HIP vs C oracle (GPU)
SHAPES = [  # (K, R, D)
    (1, 1, 1), (1, 7, 1), (3, 5, 4), (5, 6, 40), (50, 16, 40), (2, 3, 700), (4, 1, 51), (3, 9, 7), (2, 130, 8),
    (1, 1, 256), (2, 2, 260), (64, 3, 12), (1, 4096, 1), (2, 3, 2500), (1, 2, 4100), (3, 1000, 3), (7, 33, 2048),
]


def _cmp(a, b, rtol, atol):
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_allclose(a[k], b[k], rtol=rtol, atol=atol, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", SHAPES)
@pytest.mark.parametrize("kfast", [False, True])
def test_hip_logistic_sample_and_backward(hip, orc, K, R, D, kfast):
    rng = np.random.RandomState(K * 1000 + R * 10 + D)
    M = R * D
    loc = rng.standard_normal(M).astype(np.float32)
    sc = np.exp(0.5 * rng.standard_normal(M)).astype(np.float32)
    u = rng.uniform(1e-6, 1 - 1e-6, K * M).astype(np.float32)
    a, b = hip.logistic_sample(loc, sc, u, K, D, kfast=kfast), orc.logistic_sample(loc, sc, u, K, D, kfast=kfast)
    np.testing.assert_allclose(a["z"], b["z"], rtol=1e-5, atol=2e-6 * float(sc.max()))
    np.testing.assert_allclose(a["lp"], b["lp"], rtol=1e-5, atol=4e-6 * max(1, D))
    a0 = hip.logistic_sample(loc, sc, u, K, D, kfast=kfast, want_lp=False)
    assert np.array_equal(a0["z"], a["z"])
    gz = rng.standard_normal(K * M).astype(np.float32)
    glp = rng.standard_normal(K * R).astype(np.float32)
    _cmp(hip.logistic_sample_bwd(sc, u, gz, glp, K, D), orc.logistic_sample_bwd(sc, u, gz, glp, K, D), 1e-4, 1e-4 * np.sqrt(K))
    _cmp(hip.logistic_sample_bwd(sc, u, None, glp, K, D), orc.logistic_sample_bwd(sc, u, None, glp, K, D), 1e-4, 1e-4)
    _cmp(hip.logistic_sample_bwd(sc, u, gz, None, K, D), orc.logistic_sample_bwd(sc, u, gz, None, K, D), 1e-4, 1e-4 * np.sqrt(K))
    # in-kernel Philox draw: same (seed, offset) -> same u on both implementations, forward and backward
    a, b = hip.logistic_sample(loc, sc, None, K, D, seed=77, off=5), orc.logistic_sample(loc, sc, None, K, D, seed=77, off=5)
    np.testing.assert_allclose(a["z"], b["z"], rtol=1e-5, atol=4e-6 * float(sc.max()))
    np.testing.assert_allclose(a["lp"], b["lp"], rtol=1e-5, atol=4e-6 * max(1, D))
    _cmp(hip.logistic_sample_bwd(sc, None, gz, glp, K, D, 77, 5), orc.logistic_sample_bwd(sc, None, gz, glp, K, D, 77, 5),
         2e-4, 2e-4 * np.sqrt(K))


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", SHAPES)
def test_hip_logistic_uniform_logprob_periods(hip, orc, K, R, D):
    rng = np.random.RandomState(7 + K + R + D)
    N = K * R * D
    full = lambda: rng.standard_normal(N).astype(np.float32)
    glp = rng.standard_normal(K * R).astype(np.float32)
    combos = [(N, N, N), (N, R * D, R * D), (N, R * D, 1), (R * D, N, 1), (N, 1, 1)]
    for Px, Pm, Ps in combos:
        x, loc = (3 * full())[:Px].copy(), full()[:Pm].copy()
        sc = np.exp(0.3 * full()[:Ps]).astype(np.float32)
        for kfast in (False, True):
            a, b = hip.logistic_lp(x, loc, sc, K, R, D, kfast), orc.logistic_lp(x, loc, sc, K, R, D, kfast)
            np.testing.assert_allclose(a["lp"], b["lp"], rtol=2e-5, atol=4e-6 * max(1, D))
        _cmp(hip.logistic_lp_bwd(x, loc, sc, glp, K, R, D), orc.logistic_lp_bwd(x, loc, sc, glp, K, R, D), 1e-4, 2e-6)
        _cmp(hip.logistic_lp_bwd(x, loc, sc, glp, K, R, D, (False, True, False)),
             orc.logistic_lp_bwd(x, loc, sc, glp, K, R, D, (False, True, False)), 1e-4, 2e-6)
        # Uniform: low = loc - |.|, high = loc + |.|, x anywhere (some outside the support -> -inf on both)
        low = (loc - 1.0).astype(np.float32)
        high = (low[:1] + 2.5 + np.abs(full()[:Ps])).astype(np.float32) if Ps == 1 else None
        if high is None:
            high = (np.resize(low, Ps) + 2.0 + np.abs(full()[:Ps])).astype(np.float32)
            if Ps != Pm:
                continue
        for kfast in (False, True):
            a, b = hip.uniform_lp(x, low, high, K, R, D, kfast), orc.uniform_lp(x, low, high, K, R, D, kfast)
            assert np.array_equal(np.isinf(a["lp"]), np.isinf(b["lp"]))
            fin = np.isfinite(b["lp"])
            np.testing.assert_allclose(a["lp"][fin], b["lp"][fin], rtol=2e-5, atol=4e-6 * max(1, D))


@pytest.mark.gpu
@pytest.mark.parametrize("N,Pl,Ph", [(1, 1, 1), (7, 7, 1), (4096, 4096, 4096), (4096, 1024, 1), (1000, 250, 1000), (12, 12, 4),
                                     (1 << 20, 1 << 10, 1 << 20)])
@pytest.mark.parametrize("reparam", [True, False])
def test_hip_uniform_sample(hip, orc, N, Pl, Ph, reparam):
    rng = np.random.RandomState(N + Pl)
    low = rng.standard_normal(Pl).astype(np.float32)
    high = (low.max() + 0.1 + np.exp(rng.standard_normal(Ph))).astype(np.float32)
    u = rng.uniform(size=N).astype(np.float32)
    a, b = hip.uniform_sample(low, high, u, N, reparam), orc.uniform_sample(low, high, u, N, reparam)
    assert np.array_equal(a["out"], b["out"]) and np.array_equal(a["cache"], b["cache"])     # two roundings, bit-exact
    a, b = hip.uniform_sample(low, high, None, N, reparam, 5, 9), orc.uniform_sample(low, high, None, N, reparam, 5, 9)
    assert np.array_equal(a["out"], b["out"]) and np.array_equal(a["cache"], b["cache"])     # Philox words are integers
    a2 = hip.uniform_sample(low, high, None, N, reparam, 5, 9, want_cache=False)
    assert np.array_equal(a2["out"], a["out"])
    assert np.array_equal(hip.philox_u(N, 5, 9), orc.philox_u(N, 5, 9))
    rs = torch.tensor([5, 4], dtype=torch.int64)
    assert np.array_equal(hip.philox_u(N, 0, 5, rs.to("cuda:0")), orc.philox_u(N, 0, 5, rs))


@pytest.mark.gpu
def test_hip_locscale_empty_unaligned_f64(hip, orc, hip64, orc64):
    e = np.zeros(0, np.float32)
    for r in (hip, orc):
        assert r.logistic_sample(e, e, None, 3, 1)["z"].shape == (3, 0)
        assert r.uniform_sample(np.ones(1, np.float32), np.ones(1, np.float32), None, 0, True)["out"].shape == (0,)
    # unaligned base pointers (views one element into a buffer) must take the scalar path and agree
    rng = np.random.RandomState(3)
    K, R, D = 3, 5, 8
    M = R * D
    loc_b = torch.tensor(rng.standard_normal(M + 1).astype(np.float32), device="cuda:0")
    sc_b = torch.tensor(np.exp(0.2 * rng.standard_normal(M + 1)).astype(np.float32), device="cuda:0")
    u_b = torch.tensor(rng.uniform(0.01, 0.99, K * M + 1).astype(np.float32), device="cuda:0")
    z = torch.empty(K * M + 1, device="cuda:0")
    lp = torch.empty(K * R, device="cuda:0")
    hip.call("zs_logistic_sample_logprob_f32", loc_b[1:], sc_b[1:], u_b[1:], 0, 0, None, z[1:], lp, K, M, D, R, 1, None)
    ref = orc.logistic_sample(loc_b[1:].cpu().numpy(), sc_b[1:].cpu().numpy(), u_b[1:].cpu().numpy(), K, D)
    np.testing.assert_allclose(z[1:].cpu().numpy().reshape(K, M), ref["z"], rtol=1e-5, atol=4e-6)
    np.testing.assert_allclose(lp.cpu().numpy().reshape(K, R), ref["lp"], rtol=1e-5, atol=4e-5)
    # float64 twins
    for (K, R, D) in [(3, 5, 4), (2, 3, 2500), (1, 7, 1), (4, 1, 51)]:
        M = R * D
        loc = rng.standard_normal(M)
        sc = np.exp(0.5 * rng.standard_normal(M))
        u = rng.uniform(1e-9, 1 - 1e-9, K * M)
        a, b = hip64.logistic_sample(loc, sc, u, K, D), orc64.logistic_sample(loc, sc, u, K, D)
        _cmp(a, b, 1e-12, 1e-12 * max(1, D))
        gz, glp = rng.standard_normal(K * M), rng.standard_normal(K * R)
        _cmp(hip64.logistic_sample_bwd(sc, u, gz, glp, K, D), orc64.logistic_sample_bwd(sc, u, gz, glp, K, D), 1e-11, 1e-11)
        x = 3 * rng.standard_normal(K * M)
        _cmp(hip64.logistic_lp(x, loc, sc, K, R, D), orc64.logistic_lp(x, loc, sc, K, R, D), 1e-12, 1e-12 * D)
        _cmp(hip64.logistic_lp_bwd(x, loc, sc, glp, K, R, D), orc64.logistic_lp_bwd(x, loc, sc, glp, K, R, D), 1e-11, 1e-12)
        low, high = loc - 1.0, loc + 2.0
        a, b = hip64.uniform_sample(low, high, u, K * M, False), orc64.uniform_sample(low, high, u, K * M, False)
        assert np.array_equal(a["out"], b["out"]) and np.array_equal(a["cache"], b["cache"])
        xin = np.tile(low, K) + u * 3.0
        _cmp(hip64.uniform_lp(xin, low, high, K, R, D), orc64.uniform_lp(xin, low, high, K, R, D), 1e-12, 1e-12 * D)


# ------------------------------------------------------------------ product classes (host back-end on CPU, HIP on gpu)
def test_logistic_reference_unit_tests(dev):
    # test/distributions/test_logistic.py:21-77
    d = Logistic(0.1, 0.2, device=dev)
    assert d.loc.cpu() == torch.tensor(0.1) and d.scale.cpu() == torch.tensor(0.2) and d._dtype == torch.float32
    d = Logistic(torch.tensor([1., 2.]), torch.tensor([[1., 2.], [2., 3.]]), device=dev)
    assert d.loc.cpu().equal(torch.tensor([1., 2.])) and tuple(d.batch_shape) == (2, 2)
    with pytest.raises(TypeError, match=r"must have a dtype in"):
        Logistic(loc=2, scale=2, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError):
        Logistic(torch.ones([2, 1]), torch.ones([2, 4, 3]), device=dev)
    with pytest.raises(ValueError, match="scale less than zero"):
        Logistic([2.], [-1.], device=dev)
    assert d.is_reparameterized and d.is_continuous
    loc = torch.rand([2, 3], device=dev, requires_grad=True)
    scale = torch.rand([2, 3], device=dev).add_(0.1).requires_grad_()
    la = Logistic(loc, scale)
    s = la.sample()
    assert torch.norm(torch.log(la.prob(s)) - la.log_prob(s)) < 1e-5
    gl, gs = torch.autograd.grad(s.sum(), [loc, scale], allow_unused=True)
    assert gl is not None and gs is not None and torch.allclose(gl, torch.ones_like(gl))
    lp = Logistic([2.], [1.], device=dev).log_prob([3.])
    close(lp, stats.logistic.logpdf([3.], [2.], [1.]), 1e-5, 1e-6)
    # shape tables of test/distributions/utils.py (2-parameter families)
    for ls, ss, n, target in [([2, 3], [2, 1], 1, [2, 3]), ([1, 3], [2, 1], 2, [2, 2, 3]), ([2, 1, 5], [1, 3, 1], 3, [3, 2, 3, 5])]:
        dd = Logistic(torch.zeros(ls), torch.ones(ss), device=dev)
        assert list(dd.sample(n).shape) == target
        assert list(dd.log_prob(torch.zeros(target, device=dev)).shape) == target
    for dt in (torch.float32, torch.float64):
        dd = Logistic(torch.zeros([3], dtype=dt), torch.ones([3], dtype=dt), device=dev)
        assert dd.sample(2).dtype == dt and dd.log_prob(torch.zeros([3], dtype=dt)).dtype == dt


def test_logistic_goldens_through_product(dev):
    for c, g in cases("g_logistic_sample"):
        K = None if int(g["K"]) < 0 else int(g["K"])
        loc, sc = T(g["loc"], dev, True), T(g["scale"], dev, True)
        d = Logistic(loc, sc, group_ndims=int(g["g"]))
        with zs.inject_epsilon([g["u"]]):
            z = d.sample(K)
        lp = d.log_prob(None)
        assert tuple(z.shape) == g["z"].shape and tuple(lp.shape) == g["lp"].shape
        close(z, g["z"], 1e-5, 2e-6)
        close(lp, g["lp"], 1e-5, 1e-5)
        gl, gs = torch.autograd.grad((lp * T(g["w"], dev)).sum() + (z * T(g["wz"], dev)).sum(), [loc, sc])
        close(gl, g["gloc"], 1e-4, 2e-5)
        close(gs, g["gscale"], 1e-4, 5e-5)
        # the density of the SAME sample through the given-value kernel agrees with the fused one
        close(d.log_prob(z.detach().clone()), lp, 1e-5, 2e-5)
    for c, g in cases("g_logistic_logprob"):
        loc, sc, x = T(g["loc"], dev, True), T(g["scale"], dev, True), T(g["x"], dev, True)
        lp = Logistic(loc, sc, group_ndims=int(g["g"])).log_prob(x)
        assert tuple(lp.shape) == g["lp"].shape
        close(lp, g["lp"], 1e-5, 1e-5)
        gl, gs, gx = torch.autograd.grad((lp * T(g["w"], dev)).sum(), [loc, sc, x])
        close(gl, g["gloc"], 1e-4, 2e-5)
        close(gs, g["gscale"], 1e-4, 5e-5)
        close(gx, g["gx"], 1e-4, 2e-5)


def test_uniform_reference_unit_tests(dev):
    # test/distributions/test_uniform.py:20-85
    d = Uniform(0.1, 0.2, device=dev)
    assert d.low.cpu() == torch.tensor(0.1) and d.high.cpu() == torch.tensor(0.2) and d._dtype == torch.float32
    d = Uniform(torch.tensor([1., 2.]), torch.tensor([[1., 2.], [2., 3.]]), device=dev)
    assert d.low.cpu().equal(torch.tensor([1., 2.]))
    with pytest.raises(TypeError, match=r"must have a dtype in"):
        Uniform(2, 2, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError):
        Uniform(torch.zeros([2, 1]), torch.zeros([2, 4, 3]), device=dev)
    low = torch.rand([2, 3], device=dev, requires_grad=True)
    high = (low.detach() + torch.rand([2, 3], device=dev) + 0.1).requires_grad_()
    uni = Uniform(low, high)
    s = uni.sample()
    assert bool(((s >= low) & (s < high)).all())
    assert torch.norm(torch.log(uni.prob(s.detach())) - uni.log_prob(s.detach())) < 1e-6
    glo, ghi = torch.autograd.grad(s.sum(), [low, high], allow_unused=True)
    assert glo is not None and ghi is not None
    close(glo + ghi, torch.ones_like(glo), 1e-6, 1e-6)            # d/dlow + d/dhigh = (1-u) + u
    lp = Uniform(np.array([4.]), np.array([5.]), device=dev).log_prob([4.5])
    assert lp.dtype == torch.float64
    close(lp, stats.uniform.logpdf([4.5], [4.], [1.]), 1e-12, 1e-12)
    with pytest.raises(ValueError):
        Uniform(np.array([10.]), np.array([2.]), device=dev).log_prob([3.])      # low > high (test_uniform.py:81-82)
    with pytest.raises(ValueError, match="within the support"):
        Uniform(0., 1., device=dev).log_prob([1.5])
    prev = torch.distributions.Distribution._validate_args
    torch.distributions.Distribution.set_default_validate_args(False)
    try:
        assert float(Uniform(0., 1., device=dev).log_prob([1.5])) == -np.inf     # torch's unvalidated value
    finally:
        torch.distributions.Distribution.set_default_validate_args(prev)
    for ls, hs, n, target in [([2, 3], [2, 1], 1, [2, 3]), ([1, 3], [2, 1], 2, [2, 2, 3]), ([2, 1, 5], [1, 3, 1], 3, [3, 2, 3, 5])]:
        for rep in (True, False):
            dd = Uniform(torch.zeros(ls), torch.ones(hs), is_reparameterized=rep, device=dev)
            assert list(dd.sample(n).shape) == target
            assert list(dd.log_prob(torch.full(target, 0.5, device=dev)).shape) == target


def test_uniform_goldens_through_product(dev):
    for c, g in cases("g_uniform_sample"):
        K = None if int(g["K"]) < 0 else int(g["K"])
        low, high = T(g["low"], dev, True), T(g["high"], dev, True)
        d = Uniform(low, high, is_reparameterized=bool(g["reparam"]))
        with zs.inject_epsilon([g["u"]]):
            z = d.sample(K)
        assert np.array_equal(z.detach().cpu().numpy(), g["z"])
        assert np.array_equal(d.sample_cache.cpu().numpy(), g["cache"])           # uniform.py:69
        glo, ghi = torch.autograd.grad((z * T(g["wz"], dev)).sum(), [low, high])
        close(glo, g["glow"], 1e-5, 1e-5)
        close(ghi, g["ghigh"], 1e-5, 1e-5)
    g = load_golden("g_uniform_sample")
    d = Uniform(T(g["bc_low"], dev), T(g["bc_high"], dev))
    with zs.inject_epsilon([g["bc_u"]]):
        assert np.array_equal(d.sample(2).cpu().numpy(), g["bc_z"])
    # without injection the draw is still shared along the axes where only `high` broadcasts
    z = Uniform(torch.zeros(1, 3), torch.ones(4, 1), device=dev).sample(2)
    assert bool((z[:, :1] == z).all())
    for c, g in cases("g_uniform_logprob"):
        low, high = T(g["low"], dev, True), T(g["high"], dev, True)
        lp = Uniform(low, high, group_ndims=int(g["g"])).log_prob(T(g["x"], dev))
        assert tuple(lp.shape) == g["lp"].shape
        close(lp, g["lp"], 1e-5, 1e-5)
        glo, ghi = torch.autograd.grad((lp * T(g["w"], dev)).sum(), [low, high])
        close(glo, g["glow"], 1e-4, 1e-5)
        close(ghi, g["ghigh"], 1e-4, 1e-5)


def test_logistic_uniform_nodes_in_a_bayesian_net(dev):
    class Net(BayesianNet):
        def forward(self, observed):
            self.observe(observed)
            self.stochastic_node("Logistic", "a", loc=torch.zeros(5, 4), scale=torch.ones(5, 4), n_samples=3, reduce_sum_dims=[2])
            self.uniform("b", torch.zeros(5, 4), torch.full((5, 4), 2.0), n_samples=3, group_ndims=1)
            self.sn(Logistic(torch.zeros(4, device=self.device), torch.ones(4, device=self.device), group_ndims=1), "c")
            return self
    net = Net()
    net._device = dev
    net({"b": torch.full((3, 5, 4), 0.5, device=dev)})
    a, b, c = net.nodes["a"], net.nodes["b"], net.nodes["c"]
    assert list(a.log_prob().shape) == [3, 5] and list(b.log_prob().shape) == [3, 5] and list(c.log_prob().shape) == []
    close(b.log_prob(), np.full((3, 5), -4 * np.log(2.0)), 1e-6, 1e-6)
    za = a.dist.sample_cache
    close(a.log_prob(), stats.logistic.logpdf(za.cpu().numpy()).sum(-1), 1e-5, 1e-5)
    assert list(net.log_joint().shape) == [3, 5]
    # bn.py:336-358: the reference's BayesianNet.logistic() builds a LAPLACE node -- kept as is (a torch pass-through)
    d = net.logistic("d", torch.zeros(5, device=dev), torch.ones(5, device=dev))
    assert type(net.nodes["d"].dist).__name__ == "Laplace" and list(d.shape) == [5]


def test_logistic_uniform_draw_statistics(dev):
    torch.manual_seed(3)
    host_backend.manual_seed(3)
    d = Logistic(torch.zeros(4000, 8), torch.full((4000, 8), 2.0), device=dev)
    z = d.sample(4).double()
    assert abs(float(z.mean())) < 0.05 and abs(float(z.var()) - (2.0 * np.pi) ** 2 / 3) < 0.3
    z2 = d.sample(4).double()
    assert not torch.equal(z, z2)
    uu = Uniform(torch.full((4000, 8), -1.0), torch.full((4000, 8), 3.0), device=dev)
    s = uu.sample(4).double()
    assert float(s.min()) >= -1.0 and float(s.max()) < 3.0
    assert abs(float(s.mean()) - 1.0) < 0.02 and abs(float(s.var()) - 16.0 / 12) < 0.03
    c = uu.sample_cache
    assert float(c.min()) > 0.0 and float(c.max()) < 1.0


@pytest.mark.gpu
def test_logistic_graph_capture_draws_fresh_numbers():
    dev = torch.device("cuda:0")
    rng = zs.DeviceRNG(dev, seed=11)
    loc = torch.zeros(64, 40, device=dev, requires_grad=True)
    scale = torch.ones(64, 40, device=dev, requires_grad=True)
    with zs.device_rng(rng):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                rng.begin_step()
                d = Logistic(loc, scale, group_ndims=1)
                (d.sample(5).sum() + d.log_prob(None).sum()).backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        loc.grad = None
        scale.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            rng.begin_step()
            d = Logistic(loc, scale, group_ndims=1)      # scale > 0 check is skipped during capture
            z = d.sample(5)
            lp = d.log_prob(None)
            (z.sum() + lp.sum()).backward()
        g.replay()
        torch.cuda.synchronize()
        z1, g1 = z.clone(), scale.grad.clone()
        g.replay()
        torch.cuda.synchronize()
        assert not torch.equal(z1, z) and not torch.equal(g1, scale.grad)
        close(lp, stats.logistic.logpdf(z.detach().cpu().numpy()).sum(-1), 1e-5, 2e-5)


# ------------------------------------------------------------------ L2 backward reduced over the K particles
def _ksum_case(K, R, D, seed=0):
    rng = np.random.RandomState(seed + K * 100 + R * 10 + D)
    M = R * D
    x = (3 * rng.standard_normal(K * M)).astype(np.float32)
    loc = rng.standard_normal(M).astype(np.float32)
    sc = np.exp(0.3 * rng.standard_normal(M)).astype(np.float32)
    glp = rng.standard_normal(K * R).astype(np.float32)
    return x, loc, sc, glp


def test_c_oracle_logistic_bwd_ksum_is_the_folded_elementwise_backward(orc, orc64):
    """The K-summed backward equals the element-wise partials (pinned above against the reference's autograd) folded over
    the K repetitions of the parameters; gx is the same tensor."""
    for raw, tol in ((orc, 2e-6), (orc64, 1e-13)):
        for (K, R, D) in [(5, 3, 4), (2, 1, 1), (7, 2, 6), (1, 4, 3)]:
            x, loc, sc, glp = _ksum_case(K, R, D)
            M = R * D
            full = raw.logistic_lp_bwd(x, np.tile(loc, K), np.tile(sc, K), glp, K, R, D)
            for kfast in (False, True):
                got = raw.logistic_lp_bwd_ksum(x, loc, sc, glp, K, R, D, kfast=kfast)
                np.testing.assert_allclose(got["gx"], full["gx"], rtol=tol, atol=tol)
                np.testing.assert_allclose(got["gloc"], full["gloc"].reshape(K, M).sum(0), rtol=10 * tol, atol=10 * tol)
                np.testing.assert_allclose(got["gscale"], full["gscale"].reshape(K, M).sum(0), rtol=10 * tol, atol=10 * tol)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.call("zs_logistic_logprob_bwd_ksum_f32", None, orc.t(loc), orc.t(sc), orc.t(glp), 1, 1, None, orc.empty(1), orc.empty(1), 1, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", [(50, 256, 40), (5, 3, 4), (2, 1, 1), (7, 2, 6), (3, 5, 2500), (50, 64, 8), (1, 4, 3), (13, 300, 44)])
def test_hip_logistic_bwd_ksum(hip, orc, K, R, D):
    x, loc, sc, glp = _ksum_case(K, R, D)
    for kfast in (False, True):
        for want_gx in (True, False):
            a = hip.logistic_lp_bwd_ksum(x, loc, sc, glp, K, R, D, want_gx, kfast)
            b = orc.logistic_lp_bwd_ksum(x, loc, sc, glp, K, R, D, want_gx, kfast)
            _cmp(a, b, 2e-4, 2e-5 * np.sqrt(K))


@pytest.mark.gpu
def test_hip_logistic_bwd_ksum_f64(hip64, orc64):
    for (K, R, D) in [(5, 3, 4), (4, 7, 51)]:
        x, loc, sc, glp = _ksum_case(K, R, D)
        a = hip64.logistic_lp_bwd_ksum(x.astype(np.float64), loc.astype(np.float64), sc.astype(np.float64), glp.astype(np.float64), K, R, D)
        b = orc64.logistic_lp_bwd_ksum(x.astype(np.float64), loc.astype(np.float64), sc.astype(np.float64), glp.astype(np.float64), K, R, D)
        _cmp(a, b, 1e-11, 1e-12)
