"""The C ABI itself (include/zs_hip.h).

not gpu : both shared objects load and export every declared symbol; the plain-C oracle is checked
          against golden fixtures / known answers through raw ABI calls.
gpu     : libzs_hip.so vs the C oracle, entry point by entry point, on seeded inputs covering the
          vector and serial kernel paths, ragged / empty / unaligned inputs, K > 64, periodic operands
          and strided row outputs.
"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, host_kernel_library, build_oracle_lib
from zhusuan import _hip

HDR = os.path.join(ROOT, "include", "zs_hip.h")


def declared_symbols():
    src = open(HDR).read()
    return sorted(set(re.findall(r"\b(zs_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_by_both_libraries():
    names = declared_symbols()
    assert len(names) == 88 and set(_hip.PROTOTYPES) <= set(names)
    hip = ctypes.CDLL(_hip.LIB_PATH)           # loads without a GPU; no compute call is made here
    orc = ctypes.CDLL(build_oracle_lib())
    for n in names:
        assert hasattr(hip, n), "libzs_hip.so lacks %s" % n
        assert hasattr(orc, n), "libzs_oracle.so lacks %s" % n
    k = _hip.KernelLibrary(_hip.LIB_PATH)
    assert k.cdll.zs_abi_version() == _hip.ABI_VERSION == 15
    assert "release (no environment knobs)" in k.build_info() and "ABI 15" in k.build_info()
    assert b"invalid argument" in k.cdll.zs_error_string(-1)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.KernelLibrary(str(tmp_path / "nope.so"))


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10
    orc = ctypes.CDLL(build_oracle_lib())
    f = orc.zs_oracle_philox4x32_10
    f.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    f.restype = None
    out = (ctypes.c_uint32 * 4)()
    f(0, 0, 0, out)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f(0xffffffffffffffff, 0xffffffffffffffff, 0xffffffffffffffff, out)
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    f(0x85a308d3243f6a88, 0x0370734413198a2e, 0x299f31d0a4093822, out)
    assert [hex(v) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


# ------------------------------------------------------------------ raw-call helpers
class Raw(object):
    def __init__(self, klib, device, dtype=torch.float32):
        self.k, self.dev, self.dtype = klib, torch.device(device), dtype
        self.sfx = "_f32" if dtype == torch.float32 else "_f64"

    def t(self, a):
        if a is None:
            return None
        return torch.as_tensor(np.ascontiguousarray(a), dtype=self.dtype).to(self.dev)

    def empty(self, *shape):
        return torch.full(shape, float("nan"), dtype=self.dtype, device=self.dev)

    def call(self, name, *args):
        name = name.replace("_f32", self.sfx)
        conv = []
        for a in args:
            conv.append(_hip.ptr(a) if isinstance(a, torch.Tensor) or a is None else a)
        st = None
        if self.dev.type == "cuda":
            st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        self.k.call(name, *conv, st)
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    # each op returns a dict of numpy outputs
    def normal_sample(self, mu, sigma, eps, K, D, seed=0, off=0, kfast=False, want_lp=True, rs=None, ls=0, used=None):
        M = mu.size
        R = M // D
        z = self.empty(K, M)
        lp = self.empty(R, K) if kfast else self.empty(K, R)
        sk, sr = (1, K) if kfast else (R, 1)
        self.call("zs_normal_sample_logprob_f32", self.t(mu), self.t(sigma), self.t(eps), seed, off, rs, z,
                  lp if want_lp else None, K, M, D, sk, sr, ls, used)
        lpn = lp.cpu().numpy()
        return dict(z=z.cpu().numpy(), lp=lpn.T if kfast else lpn)

    def normal_sample_pair(self, mu, sigma, K, D, seed=0, off=0, rs=None, ls=0, used=None, want_lp=True):
        """Two draws of K particles (call ids off, off + 1) through the pair entry point: z [2, K, M], lp [2, K, R] (from the K-fastest
        [R, 2 K] matrix the call writes)."""
        M = mu.size
        R = M // D
        z, lp = self.empty(2 * K, M), self.empty(R, 2 * K)
        self.call("zs_normal_sample_logprob_pair_f32", self.t(mu), self.t(sigma), seed, off, rs, z, lp if want_lp else None, K, M, D,
                  1, 2 * K, ls, used)
        return dict(z=z.cpu().numpy().reshape(2, K, M), lp=lp.cpu().numpy().T.reshape(2, K, R))

    def normal_sample_bwd(self, sigma, eps, gz, glp, K, D, seed=0, off=0, rs=None, ls=0):
        M = sigma.size
        R = M // D
        gmu, gs = self.empty(M), self.empty(M)
        self.call("zs_normal_sample_logprob_bwd_f32", self.t(sigma), self.t(eps), seed, off, rs, self.t(gz), self.t(glp),
                  R, 1, gmu, gs, K, M, D, ls)
        return dict(gmu=gmu.cpu().numpy(), gsigma=gs.cpu().numpy())

    def normal_lp(self, x, mu, sigma, K, R, D, kfast=False, ls=0):
        lp = self.empty(R, K) if kfast else self.empty(K, R)
        sk, sr = (1, K) if kfast else (R, 1)
        self.call("zs_normal_logprob_f32", self.t(x), x.size, self.t(mu), mu.size, self.t(sigma), sigma.size, lp,
                  K, R, D, sk, sr, ls)
        lpn = lp.cpu().numpy()
        return dict(lp=lpn.T if kfast else lpn)

    def normal_lp_bwd(self, x, mu, sigma, glp, K, R, D, ls=0):
        N = K * R * D
        gx, gm, gs = self.empty(N), self.empty(N), self.empty(N)
        self.call("zs_normal_logprob_bwd_f32", self.t(x), x.size, self.t(mu), mu.size, self.t(sigma), sigma.size,
                  self.t(glp), R, 1, gx, gm, gs, K, R, D, ls)
        return dict(gx=gx.cpu().numpy(), gmu=gm.cpu().numpy(), gsigma=gs.cpu().numpy())

    def normal_lp_bwd_ksum(self, x, mu, sigma, glp, K, R, D, want_gx=True, ls=0):
        gx = self.empty(K * R * D)
        gm, gs = self.empty(R * D), self.empty(R * D)
        self.call("zs_normal_logprob_bwd_ksum_f32", self.t(x), self.t(mu), self.t(sigma), self.t(glp), R, 1,
                  gx if want_gx else None, gm, gs, K, R, D, ls)
        out = dict(gmu=gm.cpu().numpy(), gsigma=gs.cpu().numpy())
        if want_gx:
            out["gx"] = gx.cpu().numpy()
        return out

    def bern_lp(self, p, x, K, R, D, logits=False, kfast=False, want_p=False):
        lp = self.empty(R, K) if kfast else self.empty(K, R)
        sk, sr = (1, K) if kfast else (R, 1)
        po = self.empty(K * R * D)
        if logits:
            self.call("zs_bernoulli_logits_logprob_f32", self.t(p), self.t(x), x.size, lp, po if want_p else None,
                      K, R, D, sk, sr)
        else:
            self.call("zs_bernoulli_logprob_f32", self.t(p), self.t(x), x.size, lp, K, R, D, sk, sr)
        lpn = lp.cpu().numpy()
        out = dict(lp=lpn.T if kfast else lpn)
        if want_p:
            out["p"] = po.cpu().numpy()
        return out

    def bern_lp_bwd(self, p, x, glp, K, R, D, logits=False):
        gp = self.empty(K * R * D)
        name = "zs_bernoulli_logits_logprob_bwd_f32" if logits else "zs_bernoulli_logprob_bwd_f32"
        self.call(name, self.t(p), self.t(x), x.size, self.t(glp), R, 1, gp, K, R, D)
        return dict(gp=gp.cpu().numpy())

    def bern_lp_bwd_x(self, p, glp, K, R, D, Px, logits=False, kfast=False, gscale=None):
        """zs_bernoulli_logprob_bwd_x: the gradient w.r.t. an observation of period Px; glp [K, R] (row-major, or K-fastest)."""
        gx = self.empty(Px)
        g2 = np.ascontiguousarray(glp.reshape(K, R).T) if kfast else glp
        self.call("zs_bernoulli_logprob_bwd_x_f32", self.t(p), int(logits), Px, self.t(g2), 1 if kfast else R, K if kfast else 1,
                  self.t(gscale), 0 if gscale is None else 1, gx, K, R, D)
        return gx.cpu().numpy()

    def iw(self, logp, logq, est):
        B, K = logp.shape
        cost, bound = self.empty(B), self.empty(B)
        cp, cq = self.empty(B, K), self.empty(B, K)
        self.call("zs_iw_reduce_f32", self.t(logp), K, self.t(logq), K, B, K, est, cost, bound, cp, cq)
        return dict(cost=cost.cpu().numpy(), bound=bound.cpu().numpy(), cp=cp.cpu().numpy(), cq=cq.cpu().numpy())

    def lme(self, x):
        B, K = x.shape
        out = self.empty(B)
        self.call("zs_log_mean_exp_f32", self.t(x), K, B, K, out)
        return out.cpu().numpy()

    def philox(self, n, seed, off, rs=None):
        out = self.empty(n)
        self.call("zs_philox_normal_f32", out, n, seed, off, rs)
        return out.cpu().numpy()

    def bern_sample(self, p, n, seed, off, rs=None):
        out = self.empty(n)
        self.call("zs_bernoulli_sample_f32", self.t(p), p.size, out, n, seed, off, rs)
        return out.cpu().numpy()


@pytest.fixture(scope="module")
def orc():
    return Raw(host_kernel_library(), "cpu")


@pytest.fixture(scope="module")
def hip():
    return Raw(_hip.KernelLibrary(_hip.LIB_PATH), "cuda:0")


@pytest.fixture(scope="module")
def orc64():
    return Raw(host_kernel_library(), "cpu", torch.float64)


@pytest.fixture(scope="module")
def hip64():
    return Raw(_hip.KernelLibrary(_hip.LIB_PATH), "cuda:0", torch.float64)


# ------------------------------------------------------------------ C oracle vs goldens (CPU)
def test_c_oracle_normal_golden(orc):
    g = load_golden("g_normal_sample")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        K = max(int(g[p + "K"]), 1)
        mu, sd, eps = g[p + "mu"], g[p + "sd"], g[p + "eps"]
        if mu.shape != sd.shape:        # the kernels take same-shape operands; other broadcasts are expanded by the caller
            full = np.broadcast_shapes(mu.shape, sd.shape)
            eps = np.broadcast_to(eps.reshape((K,) + (1,) * (len(full) - mu.ndim) + mu.shape) if eps.ndim > mu.ndim else
                                  eps.reshape((1,) * (len(full) - mu.ndim) + mu.shape), ((K,) if eps.ndim > mu.ndim else ()) + full)
            mu, sd = np.ascontiguousarray(np.broadcast_to(mu, full)), np.ascontiguousarray(np.broadcast_to(sd, full))
            eps = np.ascontiguousarray(eps)
        D = int(np.prod(mu.shape[mu.ndim - int(g[p + "g"]):]))
        out = orc.normal_sample(mu.ravel(), sd.ravel(), eps.ravel(), K, D)
        assert np.array_equal(out["z"].reshape(g[p + "z"].shape), g[p + "z"])
        np.testing.assert_allclose(out["lp"].reshape(g[p + "lp"].shape), g[p + "lp"], rtol=1e-5, atol=1e-5)


def test_c_oracle_bernoulli_golden(orc):
    g = load_golden("g_bernoulli")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        if int(g[p + "from_logits"]):
            par, logits = g[p + "logits"], True
        else:
            par, logits = g[p + "probs"], False
        x = np.broadcast_to(g[p + "x"], np.broadcast_shapes(g[p + "x"].shape, par.shape)) if g[p + "x"].shape != par.shape and g[p + "x"].size == par.size else g[p + "x"]
        gnd = int(g[p + "g"])
        D = int(np.prod(par.shape[par.ndim - gnd:])) if gnd else 1
        rows = par.size // D
        out = orc.bern_lp(par.ravel(), np.ascontiguousarray(x).ravel(), 1, rows, D, logits=logits)
        np.testing.assert_allclose(out["lp"].reshape(g[p + "lp"].shape), g[p + "lp"], rtol=1e-5, atol=1e-4 if gnd else 2e-6)


def test_c_oracle_iw_golden(orc):
    g = load_golden("g_iw")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        lp, lq = g[p + "logp"].T.copy(), g[p + "logq"].T.copy()      # K-fastest rows [B, K]
        B = lp.shape[0]
        for est, tag in [(0, "sgvb"), (1, "vimco")]:
            out = orc.iw(lp, lq, est)
            c32, c64 = float(g[p + tag + "_cost"]), float(g[p + tag + "_cost64"])
            assert abs(out["cost"].mean() - c32) <= 3e-6 * abs(c32) + 1.5 * abs(c32 - c64)
            np.testing.assert_allclose(out["cp"].T / B, g[p + tag + "_glogp"], rtol=2e-5, atol=1e-7)
            ref_err = np.abs(g[p + tag + "_glogq"] - g[p + tag + "_glogq64"]).max()
            np.testing.assert_allclose(out["cq"].T / B, g[p + tag + "_glogq"], rtol=2e-5, atol=max(3 * ref_err, 2e-7))
            np.testing.assert_allclose(out["bound"], g[p + "bound"], rtol=2e-6, atol=1e-5)


def test_c_oracle_rejects_bad_arguments(orc):
    one = np.ones(4, np.float32)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.normal_sample(one, one, one, 1, 3)                # D does not divide M
    with pytest.raises(RuntimeError, match="code -1"):
        orc.normal_lp(np.ones(3, np.float32), one, one, 1, 1, 4)   # period does not divide N
    with pytest.raises(RuntimeError, match="code -1"):
        orc.iw(np.ones((2, 1), np.float32), np.ones((2, 1), np.float32), 1)   # vimco needs K >= 2


# ------------------------------------------------------------------ HIP vs C oracle (GPU)
def _cmp(a, b, rtol=2e-5, atol=2e-5):
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_allclose(a[k], b[k], rtol=rtol, atol=atol, err_msg=k)


NORMAL_SHAPES = [  # (K, R, D)
    (1, 1, 1), (1, 7, 1), (3, 5, 4), (5, 6, 40), (50, 16, 40), (2, 3, 700), (4, 1, 51), (3, 9, 7), (2, 130, 8),
    (1, 1, 256), (2, 2, 260), (64, 3, 12), (9, 256, 260), (10, 1, 700), (3, 70, 60), (4, 33, 28), (17, 5, 100),
]


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", NORMAL_SHAPES)
@pytest.mark.parametrize("kfast", [False, True])
def test_hip_normal_sample_and_backward(hip, orc, K, R, D, kfast):
    rng = np.random.RandomState(K * 1000 + R * 10 + D)
    M = R * D
    mu = rng.standard_normal(M).astype(np.float32)
    sd = np.exp(0.5 * rng.standard_normal(M)).astype(np.float32)
    eps = rng.standard_normal(K * M).astype(np.float32)
    a, b = hip.normal_sample(mu, sd, eps, K, D, kfast=kfast), orc.normal_sample(mu, sd, eps, K, D, kfast=kfast)
    assert np.array_equal(a["z"], b["z"]), "z = mu + sigma*eps must be bit-exact"
    np.testing.assert_allclose(a["lp"], b["lp"], rtol=1e-5, atol=1e-5 * max(1, D))
    gz = rng.standard_normal(K * M).astype(np.float32)
    glp = rng.standard_normal(K * R).astype(np.float32)
    _cmp(hip.normal_sample_bwd(sd, eps, gz, glp, K, D), orc.normal_sample_bwd(sd, eps, gz, glp, K, D), 1e-4, 1e-4)
    _cmp(hip.normal_sample_bwd(sd, eps, None, glp, K, D), orc.normal_sample_bwd(sd, eps, None, glp, K, D), 1e-4, 1e-4)
    # Philox path: same (seed, offset) -> same draw in forward and backward, on both implementations
    a, b = hip.normal_sample(mu, sd, None, K, D, seed=77, off=5), orc.normal_sample(mu, sd, None, K, D, seed=77, off=5)
    np.testing.assert_allclose(a["z"], b["z"], rtol=0, atol=2e-5 * float(sd.max()) * 6)
    _cmp(hip.normal_sample_bwd(sd, None, gz, glp, K, D, 77, 5), orc.normal_sample_bwd(sd, None, gz, glp, K, D, 77, 5),
         2e-4, 2e-4 * np.sqrt(K))
    # every combination of the optional gradients (each is its own kernel instantiation)
    _cmp(hip.normal_sample_bwd(sd, None, gz, None, K, D, 77, 5), orc.normal_sample_bwd(sd, None, gz, None, K, D, 77, 5),
         2e-4, 2e-4 * np.sqrt(K))
    _cmp(hip.normal_sample_bwd(sd, eps, gz, None, K, D), orc.normal_sample_bwd(sd, eps, gz, None, K, D), 1e-4, 1e-4)
    _cmp(hip.normal_sample_bwd(sd, None, None, glp, K, D, 77, 5), orc.normal_sample_bwd(sd, None, None, glp, K, D, 77, 5), 1e-4, 1e-4)


PAIR_SHAPES = [(50, 256, 40), (1, 512, 40), (5, 8, 40), (40, 64, 40), (7, 33, 12), (3, 5, 7), (2, 1, 784), (64, 300, 16), (10, 50, 14),
               (10, 1, 700), (10, 1, 51), (5, 40, 100), (1, 3, 100), (33, 2, 4), (2, 2600, 52)]     # long rows, wave rows, small rows


def test_c_oracle_pair_draw_is_two_draws(orc):
    rng = np.random.RandomState(3)
    for K, R, D in PAIR_SHAPES[2:6]:
        mu, sd = rng.standard_normal(R * D).astype(np.float32), np.exp(0.3 * rng.standard_normal(R * D)).astype(np.float32)
        pair = orc.normal_sample_pair(mu, sd, K, D, seed=11, off=4)
        for j in range(2):
            one = orc.normal_sample(mu, sd, None, K, D, seed=11, off=4 + j, kfast=True)
            assert np.array_equal(pair["z"][j], one["z"]) and np.array_equal(pair["lp"][j], one["lp"])


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", PAIR_SHAPES)
def test_hip_pair_draw_is_bit_for_bit_two_draws(hip, orc, K, R, D):
    """zs_normal_sample_logprob_pair: one launch (flat-plane shapes) or two (the rest) -- either way each half is exactly the single
    draw with its call id: values, row sums, the published ids, with the state by value and in device memory, sigma and log sigma."""
    rng = np.random.RandomState(K + R + D)
    M = R * D
    mu = rng.standard_normal(M).astype(np.float32)
    sd = np.exp(0.5 * rng.standard_normal(M)).astype(np.float32)
    for ls in (0, 1):
        sg = np.log(sd) if ls else sd
        for rs in (None, torch.tensor([77, 1000], dtype=torch.int64, device=hip.dev)):
            used = torch.zeros(2, dtype=torch.int64, device=hip.dev)
            pair = hip.normal_sample_pair(mu, sg, K, D, seed=77, off=5, rs=rs, ls=ls, used=used)
            assert used.tolist() == [77, 5 + (1000 if rs is not None else 0)]
            for j in range(2):
                one = hip.normal_sample(mu, sg, None, K, D, seed=77, off=5 + j, kfast=True, rs=rs, ls=ls)
                assert np.array_equal(pair["z"][j], one["z"]), (j, ls)
                assert np.array_equal(pair["lp"][j], one["lp"]), (j, ls)
    only_z = hip.normal_sample_pair(mu, sd, K, D, seed=77, off=5, want_lp=False)
    assert np.array_equal(only_z["z"], hip.normal_sample_pair(mu, sd, K, D, seed=77, off=5)["z"])
    ref = orc.normal_sample_pair(mu, sd, K, D, seed=77, off=5)
    np.testing.assert_allclose(hip.normal_sample_pair(mu, sd, K, D, seed=77, off=5)["z"], ref["z"], rtol=0, atol=2e-5 * float(sd.max()) * 6)


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", NORMAL_SHAPES)
def test_hip_normal_logprob_periods(hip, orc, K, R, D):
    rng = np.random.RandomState(7 + K + R + D)
    N = K * R * D
    full = lambda: rng.standard_normal(N).astype(np.float32)
    glp = rng.standard_normal(K * R).astype(np.float32)
    combos = [(N, N, N), (N, R * D, R * D), (N, R * D, 1), (R * D, N, 1), (N, 1, 1)]
    for Px, Pm, Ps in combos:
        x, mu = full()[:Px].copy(), full()[:Pm].copy()
        sd = np.exp(0.3 * full()[:Ps]).astype(np.float32)
        for kfast in (False, True):
            a, b = hip.normal_lp(x, mu, sd, K, R, D, kfast), orc.normal_lp(x, mu, sd, K, R, D, kfast)
            np.testing.assert_allclose(a["lp"], b["lp"], rtol=2e-5, atol=2e-5 * max(1, D))
        _cmp(hip.normal_lp_bwd(x, mu, sd, glp, K, R, D), orc.normal_lp_bwd(x, mu, sd, glp, K, R, D), 1e-4, 1e-4)
    x, mu, sd = full(), full()[:R * D].copy(), np.exp(0.3 * full()[:R * D]).astype(np.float32)
    for want_gx in (True, False):
        _cmp(hip.normal_lp_bwd_ksum(x, mu, sd, glp, K, R, D, want_gx), orc.normal_lp_bwd_ksum(x, mu, sd, glp, K, R, D, want_gx),
             2e-4, 2e-4)


def _logstd_case(K, R, D):
    rng = np.random.RandomState(31 + K * 1000 + R * 10 + D)
    M = R * D
    mu = rng.standard_normal(M).astype(np.float32)
    ls = (0.7 * rng.standard_normal(M) - 0.3).astype(np.float32)
    eps = rng.standard_normal(K * M).astype(np.float32)
    gz = rng.standard_normal(K * M).astype(np.float32)
    glp = rng.standard_normal(K * R).astype(np.float32)
    return mu, ls, eps, gz, glp


def _check_logstd_against_sigma_form(lib, K, R, D, tol):
    """sigma_is_logstd (Normal(logstd=...), normal.py:56): same results as handing over sigma = exp(logstd), with the
    chain rule d/d logstd = sigma * d/d sigma applied to the gradients."""
    mu, ls, eps, gz, glp = _logstd_case(K, R, D)
    sd = np.exp(ls.astype(np.float64)).astype(np.float32)
    M = R * D
    a, b = lib.normal_sample(mu, ls, eps, K, D, ls=1), lib.normal_sample(mu, sd, eps, K, D)
    np.testing.assert_allclose(a["z"], b["z"], rtol=tol, atol=tol)
    np.testing.assert_allclose(a["lp"], b["lp"], rtol=tol, atol=tol * max(1, D))
    a, b = lib.normal_sample_bwd(ls, eps, gz, glp, K, D, ls=1), lib.normal_sample_bwd(sd, eps, gz, glp, K, D)
    np.testing.assert_allclose(a["gmu"], b["gmu"], rtol=20 * tol, atol=20 * tol)
    np.testing.assert_allclose(a["gsigma"], b["gsigma"] * sd, rtol=20 * tol, atol=20 * tol * np.sqrt(K))
    x = (mu[None, :] + sd[None, :] * eps.reshape(K, M)).astype(np.float32).ravel()
    for Ps, sel in [(M, slice(None)), (1, slice(0, 1))]:
        a, b = lib.normal_lp(x, mu, ls[sel].copy(), K, R, D, ls=1), lib.normal_lp(x, mu, sd[sel].copy(), K, R, D)
        np.testing.assert_allclose(a["lp"], b["lp"], rtol=tol, atol=tol * max(1, D))
        a = lib.normal_lp_bwd(x, mu, ls[sel].copy(), glp, K, R, D, ls=1)
        b = lib.normal_lp_bwd(x, mu, sd[sel].copy(), glp, K, R, D)
        np.testing.assert_allclose(a["gx"], b["gx"], rtol=20 * tol, atol=20 * tol)
        sd_full = np.tile(sd, K) if Ps == M else np.full(K * M, sd[0], np.float32)
        np.testing.assert_allclose(a["gsigma"], b["gsigma"] * sd_full, rtol=20 * tol, atol=20 * tol)
    a, b = lib.normal_lp_bwd_ksum(x, mu, ls, glp, K, R, D, ls=1), lib.normal_lp_bwd_ksum(x, mu, sd, glp, K, R, D)
    np.testing.assert_allclose(a["gmu"], b["gmu"], rtol=20 * tol, atol=20 * tol * np.sqrt(K))
    np.testing.assert_allclose(a["gsigma"], b["gsigma"] * sd, rtol=20 * tol, atol=20 * tol * np.sqrt(K))
    np.testing.assert_allclose(a["gx"], b["gx"], rtol=20 * tol, atol=20 * tol)


@pytest.mark.parametrize("K,R,D", [(1, 3, 4), (5, 6, 40), (3, 9, 7)])
def test_c_oracle_normal_logstd_form(orc, K, R, D):
    _check_logstd_against_sigma_form(orc, K, R, D, 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", NORMAL_SHAPES)
def test_hip_normal_logstd_form(hip, orc, K, R, D):
    """HIP kernels with log(sigma) operands: against their own sigma form and against the C oracle's logstd form."""
    _check_logstd_against_sigma_form(hip, K, R, D, 1e-5)
    mu, ls, eps, gz, glp = _logstd_case(K, R, D)
    a, b = hip.normal_sample(mu, ls, eps, K, D, ls=1, kfast=True), orc.normal_sample(mu, ls, eps, K, D, ls=1, kfast=True)
    np.testing.assert_allclose(a["z"], b["z"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(a["lp"], b["lp"], rtol=1e-5, atol=1e-5 * max(1, D))
    _cmp(hip.normal_sample_bwd(ls, eps, gz, glp, K, D, ls=1), orc.normal_sample_bwd(ls, eps, gz, glp, K, D, ls=1), 1e-4,
         1e-4 * np.sqrt(K))
    _cmp(hip.normal_sample_bwd(ls, None, gz, glp, K, D, 77, 5, ls=1), orc.normal_sample_bwd(ls, None, gz, glp, K, D, 77, 5, ls=1),
         2e-4, 2e-4 * np.sqrt(K))


def _check_rng_used(lib, dev):
    """rng_used: the sampling call publishes the Philox ids it resolved from the live rng_state; a backward that reads
    them regenerates the forward's draw even after the live state has moved on (zhusuan.DeviceRNG.begin_step)."""
    K, R, D = 6, 5, 8
    mu, ls, eps, gz, glp = _logstd_case(K, R, D)
    sd = np.exp(ls).astype(np.float32)
    state = torch.tensor([1234, 1 << 16], dtype=torch.int64, device=dev)
    used = torch.zeros(2, dtype=torch.int64, device=dev)
    z_live = lib.normal_sample(mu, sd, None, K, D, off=3, rs=state, used=used)["z"]
    assert used.cpu().tolist() == [1234, (1 << 16) + 3]
    want = lib.normal_sample_bwd(sd, None, gz, glp, K, D, off=3, rs=state)
    state[1] += 1 << 16                                      # begin_step() between forward and backward
    stale = lib.normal_sample_bwd(sd, None, gz, glp, K, D, off=3, rs=state)
    got = lib.normal_sample_bwd(sd, None, gz, glp, K, D, off=0, rs=used)
    assert np.array_equal(got["gsigma"], want["gsigma"]) and np.array_equal(got["gmu"], want["gmu"])
    assert not np.allclose(stale["gsigma"], want["gsigma"])  # the live state no longer reproduces the draw
    z_again = lib.normal_sample(mu, sd, None, K, D, off=0, rs=used)["z"]
    assert np.array_equal(z_again, z_live)


def test_c_oracle_rng_used(orc):
    _check_rng_used(orc, "cpu")


@pytest.mark.gpu
def test_hip_rng_used(hip):
    _check_rng_used(hip, "cuda:0")


BERN_SHAPES = [(1, 1, 1), (1, 5, 784), (50, 8, 784), (3, 4, 16), (2, 3, 783), (7, 2, 100), (2, 9, 260), (1, 300, 4)]


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", BERN_SHAPES)
@pytest.mark.parametrize("logits", [False, True])
def test_hip_bernoulli(hip, orc, K, R, D, logits):
    rng = np.random.RandomState(11 + K + R + D)
    N = K * R * D
    if logits:
        p = (4 * rng.standard_normal(N)).astype(np.float32)
    else:
        p = rng.uniform(0, 1, N).astype(np.float32)
        p[:: max(N // 7, 1)] = 0.0             # the +1e-8 edge cases
        p[1:: max(N // 5, 1)] = 1.0
    glp = rng.standard_normal(K * R).astype(np.float32)
    for Px in sorted({N, R * D}):
        x = (rng.uniform(size=Px) < 0.5).astype(np.float32)
        if Px == N:
            x[::3] = 0.25                      # fractional observations are legal
        for kfast in (False, True):
            a, b = hip.bern_lp(p, x, K, R, D, logits, kfast, want_p=logits), orc.bern_lp(p, x, K, R, D, logits, kfast, want_p=logits)
            np.testing.assert_allclose(a["lp"], b["lp"], rtol=2e-5, atol=3e-5 * max(1, D / 16))
            if logits:
                np.testing.assert_allclose(a["p"], b["p"], rtol=1e-6, atol=1e-7)
        a, b = hip.bern_lp_bwd(p, x, glp, K, R, D, logits), orc.bern_lp_bwd(p, x, glp, K, R, D, logits)
        np.testing.assert_allclose(a["gp"], b["gp"], rtol=1e-4, atol=1e-4 if logits else 1e-30 + 1e-4 * np.abs(b["gp"]).max())


def _iw_truth_f64(logp, logq, est):
    """float64 evaluation of importance_weighted_objective.py:16-25,123-132,152-191 on K-fastest rows."""
    lq = torch.tensor(logq, dtype=torch.float64)
    l = torch.tensor(logp, dtype=torch.float64) - lq
    B, K = l.shape
    wt = torch.softmax(l, 1)
    bound = torch.logsumexp(l, 1) - np.log(K)
    out = dict(bound=bound.numpy(), cp=(-wt).numpy())
    cost = -(wt * l).sum(1)
    cq = wt.clone()
    if est == 1:
        sub = (l.sum(1, keepdim=True) - l) / (K - 1)
        srt, idx = torch.sort(l, 1, descending=True)
        m1, m2 = srt[:, :1], srt[:, 1:2]
        e = torch.exp(l - m1)
        S = e.sum(1, keepdim=True)
        cv = torch.log((S - e + torch.exp(sub - m1)) / K) + m1
        # the arg-max column needs the second maximum: sum over the OTHER particles relative to m2
        rows, j = torch.arange(B), idx[:, 0]
        mine = torch.zeros(B, K, dtype=torch.bool)
        mine[rows, j] = True
        s2 = torch.where(mine, torch.zeros_like(l), torch.exp(torch.where(mine, m2.expand(B, K), l) - m2)).sum(1)
        cv[rows, j] = torch.log((s2 + torch.exp(sub[rows, j] - m2[:, 0])) / K) + m2[:, 0]
        signal = bound.unsqueeze(1) - cv
        cost = cost - (lq * signal).sum(1)
        cq = wt - signal
    out["cost"], out["cq"] = cost.numpy(), cq.numpy()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("B,K", [(1, 1), (3, 2), (8, 5), (256, 50), (5, 64), (4, 65), (3, 200), (2, 1000), (1, 10000)])
@pytest.mark.parametrize("spread", [1.0, 5.0, 30.0])
def test_hip_iw_reduce(hip, orc, B, K, spread):
    _check_iw_reduce(hip, orc, B, K, spread)


@pytest.mark.gpu
@pytest.mark.parametrize("B,K", [(4100, 50), (4100, 64), (4100, 2), (4099, 17), (17000, 50), (17000, 7), (66000, 50),
                                 (66001, 13), (66000, 64)])
@pytest.mark.parametrize("spread", [1.0, 30.0])
def test_hip_iw_reduce_lane_groups(hip, orc, B, K, spread):
    """K <= 64 with thousands of datapoints runs the lane-group kernel (16 / 8 / 4 lanes per datapoint by B)."""
    _check_iw_reduce(hip, orc, B, K, spread)


def _check_iw_reduce(hip, orc, B, K, spread):
    rng = np.random.RandomState(B * 7 + K)
    logp = (-550 + spread * rng.standard_normal((B, K))).astype(np.float32)
    logq = (-50 + 0.3 * spread * rng.standard_normal((B, K))).astype(np.float32)
    if K > 2:
        logp[0, 1] = logp[0, 0] = logp[0].max() + 1.0      # tie at the maximum
    for est in (0, 1):
        if est == 1 and K < 2:
            with pytest.raises(RuntimeError, match="code -1"):
                hip.iw(logp, logq, est)
            continue
        a, b = hip.iw(logp, logq, est), orc.iw(logp, logq, est)
        t = _iw_truth_f64(logp, logq, est)
        np.testing.assert_allclose(a["bound"], t["bound"], rtol=2e-6, atol=2e-5)
        np.testing.assert_allclose(a["cp"], t["cp"], rtol=1e-4, atol=1e-6)
        # The fp32 oracle (reference op order) loses accuracy as K grows: the learning signal is a
        # difference of two ~550-sized log-mean-exps.  The kernel forms it as log(S) - log(S') and must be
        # at least as close to the float64 truth as the fp32 reference math is.
        for key, floor in (("cost", 2e-5 * np.abs(t["cost"]).max()), ("cq", 2e-6)):
            err_hip = np.abs(a[key] - t[key]).max()
            err_orc = np.abs(b[key] - t[key]).max()
            assert err_hip <= max(1.5 * err_orc, floor), (key, err_hip, err_orc)
        # and agree with the fp32 oracle within the oracle's own distance to the truth
        for key in ("cost", "cq"):
            slack = 2.5 * np.abs(b[key] - t[key]).max() + 1e-5 * max(1.0, np.abs(t[key]).max())
            assert np.abs(a[key] - b[key]).max() <= slack, (key, np.abs(a[key] - b[key]).max(), slack)
    np.testing.assert_allclose(hip.lme(logp), orc.lme(logp), rtol=2e-6, atol=2e-5)


def _check_device_rng_state(raw):
    """rng_state = {seed, base}: the kernel must behave exactly like (seed, base + offset) passed by value."""
    st = torch.tensor([1234, 9], dtype=torch.int64, device=raw.dev)
    assert np.array_equal(raw.philox(1000, 0, 3, rs=st), raw.philox(1000, 1234, 12))
    p = np.linspace(0.05, 0.95, 5).astype(np.float32)
    assert np.array_equal(raw.bern_sample(p, 777, 0, 1, rs=st), raw.bern_sample(p, 777, 1234, 10))
    rng = np.random.RandomState(0)
    mu, sd = rng.standard_normal(80).astype(np.float32), np.exp(rng.standard_normal(80)).astype(np.float32)
    a = raw.normal_sample(mu, sd, None, 3, 40, seed=0, off=2, rs=st)
    b = raw.normal_sample(mu, sd, None, 3, 40, seed=1234, off=11)
    assert np.array_equal(a["z"], b["z"]) and np.array_equal(a["lp"], b["lp"])
    gz, glp = rng.standard_normal(240).astype(np.float32), rng.standard_normal(6).astype(np.float32)
    ga = raw.normal_sample_bwd(sd, None, gz, glp, 3, 40, seed=0, off=2, rs=st)
    gb = raw.normal_sample_bwd(sd, None, gz, glp, 3, 40, seed=1234, off=11)
    assert np.array_equal(ga["gsigma"], gb["gsigma"])
    st[1] += 1                                   # what DeviceRNG.begin_step does between graph replays
    assert not np.array_equal(raw.philox(1000, 0, 3, rs=st), raw.philox(1000, 1234, 12))
    assert np.array_equal(raw.philox(1000, 0, 3, rs=st), raw.philox(1000, 1234, 13))


def test_c_oracle_device_rng_state(orc):
    _check_device_rng_state(orc)


@pytest.mark.gpu
def test_hip_device_rng_state(hip):
    _check_device_rng_state(hip)


@pytest.mark.gpu
def test_hip_rng(hip, orc):
    for n in (1, 3, 4, 5, 1023, 4096 + 2):
        a, b = hip.philox(n, 1234, 9), orc.philox(n, 1234, 9)
        np.testing.assert_allclose(a, b, rtol=0, atol=3e-5)
        p = np.linspace(0.01, 0.99, 7).astype(np.float32)
        assert np.array_equal(hip.bern_sample(p, n, 5, 2), orc.bern_sample(p, n, 5, 2))   # same uniforms bit-for-bit
    big = hip.philox(1 << 20, 1, 0)
    assert abs(big.mean()) < 4e-3 and abs(big.std() - 1) < 4e-3
    assert not np.array_equal(hip.philox(64, 1, 0), hip.philox(64, 1, 1))
    assert not np.array_equal(hip.philox(64, 1, 0), hip.philox(64, 2, 0))


@pytest.mark.gpu
def test_hip_empty_and_unaligned(hip, orc):
    z = np.zeros(0, np.float32)
    hip.call("zs_normal_sample_logprob_f32", hip.t(z), hip.t(z), None, 0, 0, None, hip.empty(0), hip.empty(0), 3, 0, 1, 1, 1, 0, None)
    hip.call("zs_bernoulli_logprob_f32", hip.t(z), hip.t(np.ones(1, np.float32)), 1, hip.empty(0), 1, 0, 4, 1, 1)
    hip.call("zs_iw_reduce_f32", hip.t(np.ones(4, np.float32)), 4, hip.t(np.ones(4, np.float32)), 4, 0, 4, 0, None, None, None, None)
    with pytest.raises(RuntimeError, match="code -1"):
        hip.call("zs_normal_sample_logprob_f32", None, None, None, 0, 0, None, None, None, 1, 4, 4, 1, 1, 0, None)
    # operands offset by one float: the vector path must not be taken on misaligned pointers
    rng = np.random.RandomState(5)
    K, R, D = 3, 4, 8
    N = K * R * D
    pbuf = hip.t(np.concatenate([[0.5], rng.uniform(0.01, 0.99, N)]).astype(np.float32))
    xbuf = hip.t(np.concatenate([[0.0], (rng.uniform(size=R * D) < 0.5)]).astype(np.float32))
    lp = hip.empty(K, R)
    hip.call("zs_bernoulli_logprob_f32", pbuf[1:], xbuf[1:], R * D, lp, K, R, D, R, 1)
    ref = orc.bern_lp(pbuf[1:].cpu().numpy(), xbuf[1:].cpu().numpy(), K, R, D)["lp"]
    np.testing.assert_allclose(lp.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


def test_c_oracle_f64_matches_scipy(orc64):
    from scipy import stats
    rng = np.random.RandomState(3)
    K, R, D = 3, 4, 6
    mu, sd, eps = rng.standard_normal(R * D), np.exp(0.3 * rng.standard_normal(R * D)), rng.standard_normal(K * R * D)
    out = orc64.normal_sample(mu, sd, eps, K, D)
    z = (mu + sd * eps.reshape(K, -1)).reshape(K, R * D)
    assert out["z"].dtype == np.float64 and np.array_equal(out["z"], z)
    np.testing.assert_allclose(out["lp"], stats.norm.logpdf(z, mu, sd).reshape(K, R, D).sum(-1), rtol=1e-13, atol=1e-13)


@pytest.mark.gpu
def test_hip_f64_entry_points(hip64, orc64):
    rng = np.random.RandomState(64)
    for (K, R, D) in [(1, 1, 1), (3, 5, 4), (5, 6, 40), (2, 3, 51)]:
        M, N = R * D, K * R * D
        mu, sd = rng.standard_normal(M), np.exp(0.4 * rng.standard_normal(M))
        eps, gz, glp = rng.standard_normal(N), rng.standard_normal(N), rng.standard_normal(K * R)
        for kfast in (False, True):
            a, b = hip64.normal_sample(mu, sd, eps, K, D, kfast=kfast), orc64.normal_sample(mu, sd, eps, K, D, kfast=kfast)
            assert np.array_equal(a["z"], b["z"])
            np.testing.assert_allclose(a["lp"], b["lp"], rtol=1e-12, atol=1e-12)
        _cmp(hip64.normal_sample_bwd(sd, eps, gz, glp, K, D), orc64.normal_sample_bwd(sd, eps, gz, glp, K, D), 1e-11, 1e-11)
        a, b = hip64.normal_sample(mu, sd, None, K, D, seed=9, off=4), orc64.normal_sample(mu, sd, None, K, D, seed=9, off=4)
        np.testing.assert_allclose(a["z"], b["z"], rtol=0, atol=3e-4 * float(sd.max()))
        x = rng.standard_normal(N)
        for (Px, Pm, Ps) in [(N, N, N), (N, M, M), (N, M, 1), (M, N, 1)]:
            xx, mm, ss = x[:Px].copy(), rng.standard_normal(Pm), np.exp(0.3 * rng.standard_normal(Ps))
            _cmp(hip64.normal_lp(xx, mm, ss, K, R, D, True), orc64.normal_lp(xx, mm, ss, K, R, D, True), 1e-12, 1e-12)
            _cmp(hip64.normal_lp_bwd(xx, mm, ss, glp, K, R, D), orc64.normal_lp_bwd(xx, mm, ss, glp, K, R, D), 1e-11, 1e-11)
        _cmp(hip64.normal_lp_bwd_ksum(x, mu, sd, glp, K, R, D), orc64.normal_lp_bwd_ksum(x, mu, sd, glp, K, R, D), 1e-11, 1e-11)
        p = rng.uniform(0, 1, N)
        xb = (rng.uniform(size=M) < 0.5).astype(np.float64)
        for logits in (False, True):
            pp = 4 * rng.standard_normal(N) if logits else p
            _cmp(hip64.bern_lp(pp, xb, K, R, D, logits, True, want_p=logits), orc64.bern_lp(pp, xb, K, R, D, logits, True, want_p=logits),
                 1e-12, 1e-12)
            _cmp(hip64.bern_lp_bwd(pp, xb, glp, K, R, D, logits), orc64.bern_lp_bwd(pp, xb, glp, K, R, D, logits), 1e-10, 1e-10)
    for (B, K) in [(3, 2), (8, 50), (2, 300)]:
        logp, logq = -550 + 5 * rng.standard_normal((B, K)), -50 + rng.standard_normal((B, K))
        for est in (0, 1):
            a, t = hip64.iw(logp, logq, est), _iw_truth_f64(logp, logq, est)
            for key in ("cost", "bound", "cp", "cq"):
                np.testing.assert_allclose(a[key], t[key], rtol=1e-10, atol=1e-10, err_msg=key)
        np.testing.assert_allclose(hip64.lme(logp), orc64.lme(logp), rtol=1e-13)
    pr = np.linspace(0.05, 0.95, 5)
    assert np.array_equal(hip64.bern_sample(pr, 1001, 5, 2), orc64.bern_sample(pr, 1001, 5, 2))
    np.testing.assert_allclose(hip64.philox(1001, 1, 2), orc64.philox(1001, 1, 2), rtol=0, atol=3e-5)


def test_package_has_no_host_routing():
    """The product package holds no code that routes kernel calls anywhere but libzs_hip.so: the host back-end of this
    test-suite is a set of monkeypatches that live in tests/host_backend.py.  A fresh interpreter that imports the
    package (nothing from tests/) refuses CPU tensors, and the package's sources mention neither the oracle nor a hook."""
    import subprocess
    import sys as _sys
    pkg = os.path.join(ROOT, "zhusuan-pytorch_amd")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch, zhusuan\nfrom zhusuan import _hip\nfrom zhusuan.distributions import Normal\n"
            "assert not hasattr(_hip, '_install_host_library_for_tests') and not hasattr(_hip, '_HOST_LIB')\n"
            "try:\n    Normal(mean=torch.zeros(4), std=torch.ones(4)).sample()\n    print('COMPUTED')\n"
            "except RuntimeError as e:\n    print('REFUSED', e)\n") % pkg
    r = subprocess.run([_sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout and "no CPU path" in r.stdout, r.stdout + r.stderr
    for dirpath, _, files in os.walk(os.path.join(pkg, "zhusuan")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() and "host_library" not in src.lower(), os.path.join(dirpath, f)


# ------------------------------------------------------------------ K4b: the whole IW objective in one launch
def _iw_objective(raw, a, b, q, est, want_mean, ws_len=4096):
    B, K = q.shape
    cost_b, bound = raw.empty(B), raw.empty(B)
    coef = raw.empty(2, B, K)
    mean = raw.empty(1)
    ws = torch.zeros(ws_len, dtype=raw.dtype, device=raw.dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=raw.dev)
    raw.call("zs_iw_objective_f32", raw.t(a), K, raw.t(b), K, raw.t(q), K, B, K, est, int(want_mean), cost_b, bound, coef,
             mean if want_mean else None, ws if want_mean else None, ws_len if want_mean else 0, ticket if want_mean else None)
    assert int(ticket) == 0, "the kernel must leave the ticket at zero"
    return dict(cost=cost_b.cpu().numpy(), bound=bound.cpu().numpy(), coef=coef.cpu().numpy(),
                mean=mean.cpu().numpy() if want_mean else None)


def test_c_oracle_iw_objective_is_reduce_plus_mean(orc):
    rng = np.random.RandomState(5)
    B, K = 37, 9
    a = (-500 + 3 * rng.standard_normal((B, K))).astype(np.float32)
    b = (-40 + rng.standard_normal((B, K))).astype(np.float32)
    q = (-45 + rng.standard_normal((B, K))).astype(np.float32)
    for est in (0, 1):
        ref = orc.iw((a + b).astype(np.float32), q, est)
        for want_mean in (False, True):
            for bb in (b, None):
                got = _iw_objective(orc, a if bb is not None else (a + b).astype(np.float32), bb, q, est, want_mean)
                sc = 1.0 / B if want_mean else 1.0
                np.testing.assert_array_equal(got["cost"], ref["cost"])
                np.testing.assert_array_equal(got["bound"], ref["bound"])
                np.testing.assert_allclose(got["coef"][0], ref["cp"] * sc, rtol=1e-6, atol=0)
                np.testing.assert_allclose(got["coef"][1], ref["cq"] * sc, rtol=1e-6, atol=1e-12)
                if want_mean:
                    np.testing.assert_allclose(got["mean"][0], ref["cost"].astype(np.float64).mean(), rtol=1e-6)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.call("zs_iw_objective_f32", orc.t(a), K, None, K, orc.t(q), K, B, K, 0, 1, None, None, None, None, None, 0, None)


@pytest.mark.gpu
@pytest.mark.parametrize("B,K", [(1, 2), (3, 5), (256, 50), (5, 64), (4, 65), (3, 200), (20000, 50), (5000, 70), (2, 1000)])
@pytest.mark.parametrize("est", [0, 1])
def test_hip_iw_objective(hip, orc, B, K, est):
    rng = np.random.RandomState(B + K + est)
    a = (-500 + 3 * rng.standard_normal((B, K))).astype(np.float32)
    b = (-40 + rng.standard_normal((B, K))).astype(np.float32)
    q = (-45 + 2 * rng.standard_normal((B, K))).astype(np.float32)
    base = hip.iw((a + b).astype(np.float32), q, est)                       # the two-launch form on the same device
    for want_mean in (True, False):
        got = _iw_objective(hip, a, b, q, est, want_mean)
        ref = _iw_objective(orc, a, b, q, est, want_mean)
        sc = 1.0 / B if want_mean else 1.0
        np.testing.assert_array_equal(got["cost"], base["cost"])             # same kernel body: bit-identical rows
        np.testing.assert_array_equal(got["bound"], base["bound"])
        np.testing.assert_allclose(got["coef"][0], base["cp"] * np.float32(sc), rtol=2e-7, atol=0)
        np.testing.assert_allclose(got["coef"][1], base["cq"] * np.float32(sc), rtol=2e-7, atol=1e-12)
        np.testing.assert_allclose(got["bound"], ref["bound"], rtol=2e-6, atol=1e-4)
        if want_mean:
            exact = base["cost"].astype(np.float64).mean()
            assert abs(got["mean"][0] - exact) <= 2e-6 * abs(exact)
            again = _iw_objective(hip, a, b, q, est, True)
            assert got["mean"][0] == again["mean"][0], "the batch mean must be deterministic"
    # workspace too small for the grid is rejected, not overrun
    if B >= 5000:
        with pytest.raises(RuntimeError, match="code -1"):
            _iw_objective(hip, a, b, q, est, True, ws_len=8)


@pytest.mark.gpu
def test_hip_iw_objective_f64(hip64, orc64):
    rng = np.random.RandomState(11)
    for B, K in [(7, 5), (300, 50), (3, 130)]:
        a, b = -500 + 3 * rng.standard_normal((B, K)), -40 + rng.standard_normal((B, K))
        q = -45 + 2 * rng.standard_normal((B, K))
        for est in (0, 1):
            got, ref = _iw_objective(hip64, a, b, q, est, True), _iw_objective(orc64, a, b, q, est, True)
            np.testing.assert_allclose(got["mean"], ref["mean"], rtol=1e-10)
            np.testing.assert_allclose(got["cost"], ref["cost"], rtol=1e-9, atol=1e-7)
            np.testing.assert_allclose(got["coef"], ref["coef"], rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------ S1: scalar ELBO epilogue
def _scalar_objective(raw, vecs, coefs):
    out, cv = raw.empty(1), raw.empty(6)
    args = []
    for j in range(6):
        if j < len(vecs):
            args += [raw.t(vecs[j]), int(np.asarray(vecs[j]).size), float(coefs[j])]
        else:
            args += [None, 0, 0.0]
    raw.call("zs_scalar_objective_f32", *args, out, cv)
    return float(out.cpu().numpy()[0]), cv.cpu().numpy()[:len(vecs)]


def test_c_oracle_scalar_objective(orc):
    rng = np.random.RandomState(3)
    vecs = [rng.standard_normal(n).astype(np.float32) - 90 for n in (1, 7, 512, 3000)]
    coefs = [-1.0, 1.0 / 7, -456.0 / 512, 0.25]
    val, cv = _scalar_objective(orc, vecs, coefs)
    exact = sum(c * v.astype(np.float64).sum() for c, v in zip(coefs, vecs))
    assert abs(val - exact) <= 2e-7 * abs(exact)
    np.testing.assert_allclose(cv, np.asarray(coefs, np.float32), rtol=0, atol=0)
    with pytest.raises(RuntimeError, match="code -1"):
        _scalar_objective(orc, [], [])


@pytest.mark.gpu
def test_hip_scalar_objective(hip, orc, hip64, orc64):
    rng = np.random.RandomState(4)
    for sizes in [(1,), (5, 1), (512, 512, 512), (64, 65, 1024, 1025, 100000, 3)]:
        vecs = [rng.standard_normal(n) * 3 - 50 for n in sizes]
        coefs = list(rng.standard_normal(len(sizes)))
        exact = sum(c * np.asarray(v, np.float32).astype(np.float64).sum() for c, v in zip(coefs, vecs))
        a, ca = _scalar_objective(hip, [v.astype(np.float32) for v in vecs], coefs)
        b, cb = _scalar_objective(orc, [v.astype(np.float32) for v in vecs], coefs)
        assert abs(a - exact) <= 3e-7 * max(1.0, abs(exact)) and abs(a - b) <= 3e-7 * max(1.0, abs(exact))
        np.testing.assert_array_equal(ca, cb)
        a64, _ = _scalar_objective(hip64, vecs, coefs)
        exact64 = sum(c * v.sum() for c, v in zip(coefs, vecs))
        assert abs(a64 - exact64) <= 1e-12 * max(1.0, abs(exact64))


@pytest.mark.gpu
def test_every_entry_point_has_a_timing_id(hip):
    """bench.py asks the library for the recorded launches of every compute entry point: each name must resolve."""
    for name in _hip.PROTOTYPES:
        assert hip.k.cdll.zs_prof_kernel_id(name.encode()) >= 0, name
        assert hip.k.prof_query(name)["count"] >= 0


# ------------------------------------------------------------------ A1: Adam update of up to 32 tensors, one launch
def _adam_run(raw, sizes, steps, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, gs=1.0, missing=(), misalign=False, seed=0):
    """`steps` launches over tensors of `sizes` elements (`missing`: tensors without a gradient; `misalign`: every tensor
    4 bytes off a 16-byte boundary).  Returns (params concatenated, exp_avg, exp_avg_sq)."""
    rng = np.random.RandomState(seed + int(sum(sizes)))
    n = int(sum(sizes))
    starts = (ctypes.c_int64 * (len(sizes) + 1))(*[int(x) for x in np.concatenate([[0], np.cumsum(sizes)])])

    def tensor(a):
        return raw.t(np.concatenate([[0.0], a]))[1:] if misalign else raw.t(a)
    ps = [tensor(rng.standard_normal(k)) for k in sizes]
    m, v = raw.t(np.zeros(n)), raw.t(np.zeros(n))
    step = torch.zeros(len(sizes), dtype=torch.int64, device=raw.dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=raw.dev)
    pptr = (ctypes.c_void_p * len(sizes))(*[p.data_ptr() for p in ps])
    hyper = torch.tensor([lr, b1, b2, eps], dtype=torch.float64, device=raw.dev)
    for s in range(steps):
        scale = 10.0 ** rng.uniform(-4, 1)
        gs_ = [tensor(rng.standard_normal(k) * scale) for k in sizes]
        # `missing` tensors have no gradient on the ODD steps only: their moments are non-zero when they are skipped
        gptr = (ctypes.c_void_p * len(sizes))(*[None if (i in missing and s % 2 == 1) else g.data_ptr() for i, g in enumerate(gs_)])
        if s % 3 == 2:     # hyper-parameters from device memory (the by-value ones are then ignored)
            raw.call("zs_adam_step_f32", pptr, gptr, starts, len(sizes), m, v, step, ticket, n, 7.0, 0.1, 0.2, 3.0, gs, hyper)
        else:
            raw.call("zs_adam_step_f32", pptr, gptr, starts, len(sizes), m, v, step, ticket, n, lr, b1, b2, eps, gs, None)
    want = [steps - (steps // 2 if i in missing else 0) for i in range(len(sizes))]
    assert step.cpu().tolist() == want and int(ticket.item()) == 0
    return torch.cat(ps).cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()


def test_c_oracle_adam_is_torch_adam(orc, orc64):
    """The oracle's A1 against torch.optim.Adam itself (defaults; the callers' optimizer: vae_mnist.py:104, iwae.py:141,
    bnn_vi.py:135), same gradients, 7 steps, three tensors: one never receives a gradient, one misses it on steps 2 and 5
    (``grad is None``: torch leaves such a parameter, its moments and its step count alone -- with non-zero moments)."""
    for raw, dt, tol in ((orc, torch.float32, 3e-6), (orc64, torch.float64, 1e-13)):
        rng = np.random.RandomState(3)
        sizes = [1001, 40, 7]
        ws = [torch.nn.Parameter(torch.tensor(rng.standard_normal(k), dtype=dt)) for k in sizes]
        opt = torch.optim.Adam(ws, lr=2e-3)
        ps = [raw.t(w.detach().numpy().copy()) for w in ws]
        n = sum(sizes)
        m, v = raw.t(np.zeros(n)), raw.t(np.zeros(n))
        step, ticket = torch.zeros(3, dtype=torch.int64), torch.zeros(1, dtype=torch.int32)
        starts = (ctypes.c_int64 * 4)(0, 1001, 1041, 1048)
        pptr = (ctypes.c_void_p * 3)(*[p.data_ptr() for p in ps])
        for s in range(7):
            gs = [rng.standard_normal(k) * (10.0 ** rng.uniform(-4, 1)) for k in sizes]
            skip1 = s in (2, 5)
            ws[0].grad = torch.tensor(gs[0], dtype=dt)
            ws[1].grad = None if skip1 else torch.tensor(gs[1], dtype=dt)
            opt.step()
            gt = [raw.t(g) for g in gs]
            gptr = (ctypes.c_void_p * 3)(gt[0].data_ptr(), None if skip1 else gt[1].data_ptr(), None)
            raw.call("zs_adam_step_f32", pptr, gptr, starts, 3, m, v, step, ticket, n, 2e-3, 0.9, 0.999, 1e-8, 1.0, None)
            for p, w in zip(ps, ws):
                np.testing.assert_allclose(p.numpy(), w.detach().numpy(), rtol=tol, atol=tol * 1e-2)
        assert step.tolist() == [7, 5, 0]
        assert [int(opt.state[w]["step"]) if w in opt.state and opt.state[w] else 0 for w in ws] == [7, 5, 0]
    p = orc.t(np.zeros(8))
    one = (ctypes.c_void_p * 1)(p.data_ptr())
    two = (ctypes.c_void_p * 2)(p.data_ptr(), p.data_ptr())
    ok = (ctypes.c_int64 * 2)(0, 8)
    with pytest.raises(RuntimeError, match="code -1"):
        orc.call("zs_adam_step_f32", one, one, ok, 1, p, p, step, ticket, 8, 1e-3, 1.0, 0.999, 1e-8, 1.0, None)      # beta1 = 1
    with pytest.raises(RuntimeError, match="code -1"):
        orc.call("zs_adam_step_f32", one, one, ok, 1, p, p, None, ticket, 8, 1e-3, 0.9, 0.999, 1e-8, 1.0, None)     # no step counter
    with pytest.raises(RuntimeError, match="code -1"):                                                          # does not end at n
        orc.call("zs_adam_step_f32", two, two, (ctypes.c_int64 * 3)(0, 4, 7), 2, p, p, step, ticket, 8, 1e-3, 0.9, 0.999, 1e-8, 1.0, None)
    many = (ctypes.c_void_p * 33)(*([p.data_ptr()] * 33))
    with pytest.raises(RuntimeError, match="code -2"):                     # more than ZS_ADAM_MAX_TENSORS
        orc.call("zs_adam_step_f32", many, many, (ctypes.c_int64 * 34)(*range(34)), 33, orc.t(np.zeros(33)), orc.t(np.zeros(33)),
                 step, ticket, 33, 1e-3, 0.9, 0.999, 1e-8, 1.0, None)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,missing,misalign", [
    ([1], (), False), ([3], (), False), ([4], (), False), ([1000], (), False), ([4099], (), False), ([4096], (), True),
    ([1 << 20], (), False), ([1346864], (), False),                                   # the VAE / IWAE parameter count, flat
    ([392000, 500, 250000, 500, 20000, 40, 20000, 40, 20000, 500, 250000, 500, 392000, 784], (), False),   # ... as its tensors
    ([700, 700, 51, 51, 1], (), False),                                               # the BNN's
    ([12, 500, 3, 40], (2,), False), ([400, 4096, 64], (0,), True), (list(range(1, 33)), (5, 31), False)])
def test_hip_adam(hip, orc, sizes, missing, misalign):
    """HIP A1 against the C oracle: 16-byte and element-wise paths, tensors that straddle thread groups, missing gradients,
    the step counter and the ticket after several launches (the last workgroup publishes the count)."""
    for gs in (1.0, 0.125):
        a = _adam_run(hip, sizes, 4, gs=gs, missing=missing, misalign=misalign)
        b = _adam_run(orc, sizes, 4, gs=gs, missing=missing, misalign=misalign)
        for x, y, name in zip(a, b, ("param", "exp_avg", "exp_avg_sq")):
            # (the first moment is a sum of gradients of either sign: absolute tolerance at the tensor's scale)
            np.testing.assert_allclose(x, y, rtol=2e-6, atol=3e-7 * np.abs(y).max(), err_msg=name)


@pytest.mark.gpu
def test_hip_adam_f64(hip64, orc64):
    a, b = _adam_run(hip64, [777, 12, 5], 4), _adam_run(orc64, [777, 12, 5], 4)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, rtol=1e-13, atol=1e-15 * np.abs(y).max())


def test_both_libraries_pair_draws_on_the_same_shapes():
    """the C oracle restates by hand the shape rule by which the sampling kernel takes both draws of a latent in one
    launch (zs_normal_sample_pair_one_launch; csrc/zs_normal.hip derives it from k1_tile).  If the two predicates drifted apart,
    the host and GPU back-ends would pair draws -- and hand out Philox call ids among several latents -- differently.  A host
    function of both libraries: compared over a grid of shapes, no GPU needed."""
    hip_lib = _hip.KernelLibrary(_hip.LIB_PATH).cdll
    orc_lib = host_kernel_library().cdll
    for f in (hip_lib, orc_lib):
        f.zs_normal_sample_pair_one_launch.restype = ctypes.c_int
        f.zs_normal_sample_pair_one_launch.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int]
    n = paired = 0
    for K in (1, 2, 5, 10, 40, 50, 64, 65, 512, 4096):
        for D in (1, 2, 3, 4, 8, 12, 13, 14, 40, 51, 64, 100, 256, 784, 1024, 4096):
            for R in (1, 3, 16, 50, 64, 256, 257, 512, 4096, 100000):
                for want_lp in (0, 1):
                    a = hip_lib.zs_normal_sample_pair_one_launch(K, R * D, D, want_lp)
                    b = orc_lib.zs_normal_sample_pair_one_launch(K, R * D, D, want_lp)
                    assert a == b, ("K=%d R=%d D=%d want_lp=%d: libzs_hip says %d, the oracle %d" % (K, R, D, want_lp, a, b))
                    n += 1
                    paired += a == 1
    assert n == 10 * 16 * 10 * 2 and 0 < paired < n          # (both answers occur: the grid crosses the rule's boundary)


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D", [(9, 49157, 40), (3, 70001, 40), (50, 12300, 160)])
def test_hip_given_value_logprob_beyond_a_tile_per_resident_workgroup(hip, K, R, D):
    """The given-value kernels (K2 / L2 / U2) at sizes with more parameter-plane tiles than resident workgroups (R * D / 4 >= 1536
    tiles of 320 or 640 lanes; several chunks of particles per tile, ragged last tile): every row sum against a float64 evaluation.
    (Written for round 5's equal-contiguous-shares form of the kernel, which passed it and was measured slower -- docs/history.md;
    the parity tests of the suite otherwise stop at sizes the serial oracle finishes in seconds.)"""
    rng = np.random.RandomState(K + R)
    M = R * D
    x = rng.standard_normal((K, R, D)).astype(np.float32)
    mu = rng.standard_normal((R, D)).astype(np.float32)
    sd = np.exp(0.3 * rng.standard_normal((R, D))).astype(np.float32)
    for kfast in (True, False):
        got = hip.normal_lp(x.reshape(-1), mu.reshape(-1), sd.reshape(-1), K, R, D, kfast)["lp"].reshape(K, R)
        want = (-0.9189385332046727 - np.log(sd.astype(np.float64)) - 0.5 * ((x.astype(np.float64) - mu) / sd.astype(np.float64)) ** 2).sum(-1)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * D)
    # Logistic and Uniform, same kernel (L2, U2)
    import scipy.stats as st
    lp = hip.empty(K, R)
    hip.call("zs_logistic_logprob_f32", hip.t(x.reshape(-1)), x.size, hip.t(mu.reshape(-1)), M, hip.t(sd.reshape(-1)), M, lp, K, R, D, R, 1)
    np.testing.assert_allclose(lp.cpu().numpy(), st.logistic.logpdf(x.astype(np.float64), mu, sd).sum(-1), rtol=2e-5, atol=2e-5 * D)
    low, high = (mu - 3.0 * sd - 6.0).astype(np.float32), (mu + 3.0 * sd + 6.0).astype(np.float32)      # (every x inside the support)
    hip.call("zs_uniform_logprob_f32", hip.t(x.reshape(-1)), x.size, hip.t(low.reshape(-1)), M, hip.t(high.reshape(-1)), M, lp, K, R, D, R, 1)
    inside = (x >= low) & (x < high)
    want = np.where(inside, -np.log(high.astype(np.float64) - low), -np.inf).sum(-1)
    got = lp.cpu().numpy()
    np.testing.assert_array_equal(np.isfinite(got), np.isfinite(want))
    np.testing.assert_allclose(got[np.isfinite(want)], want[np.isfinite(want)], rtol=2e-5, atol=2e-5 * D)
