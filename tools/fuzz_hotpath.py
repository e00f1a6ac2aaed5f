#!/usr/bin/env python
"""Randomised differential run of the HOT-PATH kernels through the C ABI -- K1 (fused Normal sample + log-density, given eps and
in-kernel Philox) and its backward, K2 (Normal log-density of a value, every broadcast period) and its backwards, K3 (Bernoulli
log-mass, probs and logits) and its backward, K4 (importance-weighted reduction, sgvb and vimco), log-mean-exp, K5 (Bernoulli
sampler), Philox, the two-draws-in-one-launch mode of K1, IW1 (the generator side of the importance-weighted objective in one launch, both directions) -- libzs_hip.so on the GPU against the C oracle on the host, random shapes, for a given number of seconds.  Uses
the raw-call helpers of tests/test_cabi.py and its tolerances.  Exit code 1 at the first mismatch.

  python tools/fuzz_hotpath.py [seconds=120] [seed=0]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

from zhusuan import _hip
from conftest import host_kernel_library
from test_cabi import Raw, _iw_truth_f64
import test_iw_fused as iwf

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.RandomState(seed)
hip = Raw(_hip.KernelLibrary(_hip.LIB_PATH), "cuda:0")
orc = Raw(host_kernel_library(), "cpu")
counts = {}


def close(a, b, rtol, atol, what, shape):
    if not np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True):
        d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
        print("MISMATCH %s at %s: max abs diff %.3e (atol %.1e, rtol %.1e)" % (what, shape, np.nanmax(d), atol, rtol), flush=True)
        sys.exit(1)


def dim(hi, small=0.3):
    return int(rng.randint(1, (min(hi, 9) if rng.rand() < small else hi) + 1))


def shape3():
    K, R, D = dim(64), dim(300), (dim(64) if rng.rand() < 0.6 else dim(800))
    while K * R * D > 400000:
        R = max(R // 2, 1)
    return K, R, D


def case_k1():
    K, R, D = shape3()
    M, ls, kfast = R * D, int(rng.rand() < 0.4), bool(rng.rand() < 0.5)
    mu = rng.standard_normal(M).astype(np.float32)
    sg = (rng.uniform(-0.5, 0.3, M) if ls else rng.uniform(0.5, 1.5, M)).astype(np.float32)
    shape = (K, R, D, ls, kfast)
    if rng.rand() < 0.5:                      # eps given: z bit-exact (two roundings, like the reference)
        eps = rng.standard_normal(K * M).astype(np.float32)
        a, b = hip.normal_sample(mu, sg, eps, K, D, kfast=kfast, ls=ls), orc.normal_sample(mu, sg, eps, K, D, kfast=kfast, ls=ls)
        if ls:                                # sigma = exp(log-std): the device's expf against glibc's (1 ulp)
            close(a["z"], b["z"], 3e-7, 3e-7 * max(np.abs(b["z"]).max(), 1), "K1 z (eps given, log-std)", shape)
        elif not np.array_equal(a["z"], b["z"]):
            print("MISMATCH K1 z (eps given) not bit-exact at %s" % (shape,), flush=True)
            sys.exit(1)
    else:                                     # in-kernel Philox: device sin / cos / log against libm
        eps = None
        s, off = int(rng.randint(1 << 30)), int(rng.randint(1 << 20))
        a = hip.normal_sample(mu, sg, None, K, D, seed=s, off=off, kfast=kfast, ls=ls)
        b = orc.normal_sample(mu, sg, None, K, D, seed=s, off=off, kfast=kfast, ls=ls)
        close(a["z"], b["z"], 0, 3e-5 * max(np.abs(b["z"]).max(), 1), "K1 z (Philox)", shape)
    close(a["lp"], b["lp"], 3e-5, 3e-4 * max(D, 1) ** 0.5, "K1 lp", shape)
    gz, glp = rng.standard_normal(K * M).astype(np.float32), rng.standard_normal(K * R).astype(np.float32)
    if eps is not None:
        ga, gb = hip.normal_sample_bwd(sg, eps, gz, glp, K, D, ls=ls), orc.normal_sample_bwd(sg, eps, gz, glp, K, D, ls=ls)
        for k in ("gmu", "gsigma"):
            close(ga[k], gb[k], 2e-4, 2e-4 * max(np.abs(gb[k]).max(), 1), "K1 bwd " + k, shape)


def case_pair():
    """Both draws of a latent in one call (zs_normal_sample_logprob_pair): each half bit for bit the single draw with its call id -- the
    flat-plane kernel's two-draw mode on the shapes it takes, two launches on the rest."""
    K = int(rng.randint(1, 65))
    D = 4 * int(rng.randint(1, 33)) if rng.rand() < 0.8 else dim(60)
    R = dim(600)
    while K * R * D > 400000:
        R = max(R // 2, 1)
    M, ls = R * D, int(rng.rand() < 0.4)
    mu = rng.standard_normal(M).astype(np.float32)
    sg = (rng.uniform(-0.5, 0.3, M) if ls else rng.uniform(0.5, 1.5, M)).astype(np.float32)
    s, off = int(rng.randint(1 << 30)), int(rng.randint(1 << 20))
    pair = hip.normal_sample_pair(mu, sg, K, D, seed=s, off=off, ls=ls)
    for j in range(2):
        one = hip.normal_sample(mu, sg, None, K, D, seed=s, off=off + j, kfast=True, ls=ls)
        if not (np.array_equal(pair["z"][j], one["z"]) and np.array_equal(pair["lp"][j], one["lp"])):
            print("MISMATCH pair draw %d differs from the single draw at %s" % (j, (K, R, D, ls, s, off)), flush=True)
            sys.exit(1)


def case_k2():
    K, R, D = shape3()
    N, ls = K * R * D, int(rng.rand() < 0.4)
    per = lambda: [1, D, R * D, N][int(rng.randint(4))] if rng.rand() < 0.8 else R * D
    px, pm, ps = (N if rng.rand() < 0.7 else R * D), per(), per()
    x, mu = rng.standard_normal(px).astype(np.float32), rng.standard_normal(pm).astype(np.float32)
    sg = (rng.uniform(-0.5, 0.3, ps) if ls else rng.uniform(0.5, 1.5, ps)).astype(np.float32)
    shape = (K, R, D, px, pm, ps, ls)
    kfast = bool(rng.rand() < 0.5)
    a, b = hip.normal_lp(x, mu, sg, K, R, D, kfast=kfast, ls=ls), orc.normal_lp(x, mu, sg, K, R, D, kfast=kfast, ls=ls)
    close(a["lp"], b["lp"], 3e-5, 3e-4 * max(D, 1) ** 0.5, "K2 lp", shape)
    glp = rng.standard_normal(K * R).astype(np.float32)
    ga, gb = hip.normal_lp_bwd(x, mu, sg, glp, K, R, D, ls=ls), orc.normal_lp_bwd(x, mu, sg, glp, K, R, D, ls=ls)
    for k in ("gx", "gmu", "gsigma"):
        close(ga[k], gb[k], 2e-4, 2e-4 * max(np.abs(gb[k]).max(), 1), "K2 bwd " + k, shape)
    if px == N and pm == R * D and ps == R * D:
        want_gx = bool(rng.rand() < 0.5)
        ga = hip.normal_lp_bwd_ksum(x, mu, sg, glp, K, R, D, want_gx=want_gx, ls=ls)
        gb = orc.normal_lp_bwd_ksum(x, mu, sg, glp, K, R, D, want_gx=want_gx, ls=ls)
        for k in gb:
            close(ga[k], gb[k], 3e-4, 3e-4 * max(np.abs(gb[k]).max(), 1), "K2 bwd-ksum " + k, shape)


def case_k3():
    K, R, D = shape3()
    logits, kfast = bool(rng.rand() < 0.4), bool(rng.rand() < 0.5)
    N = K * R * D
    # (logits within +-6: beyond, log(1 - sigmoid(l) + 1e-8) amplifies the last bit of the fp32 sigmoid -- 6e-8 against 1 - p -- by
    # orders of magnitude, in the reference's own arithmetic as much as here)
    p = (np.clip(2.0 * rng.standard_normal(N), -6, 6) if logits else rng.uniform(0, 1, N)).astype(np.float32)
    if not logits and N > 4:
        p[rng.randint(N, size=3)] = [0.0, 1.0, 1e-9]                     # the edges the +1e-8 exists for
    xr = R if rng.rand() < 0.6 else K * R
    x = (rng.uniform(0, 1, xr * D) < 0.5).astype(np.float32) if rng.rand() < 0.7 else rng.uniform(0, 1, xr * D).astype(np.float32)
    shape = (K, R, D, xr, logits, kfast)
    want_p = bool(logits and rng.rand() < 0.5)
    a = hip.bern_lp(p, x, K, R, D, logits=logits, kfast=kfast, want_p=want_p)
    b = orc.bern_lp(p, x, K, R, D, logits=logits, kfast=kfast, want_p=want_p)
    close(a["lp"], b["lp"], 3e-5, (1e-4 if logits else 3e-5) * D, "K3 lp", shape)
    if want_p:
        close(a["p"], b["p"], 3e-6, 3e-7, "K3 probs_out", shape)
    glp = rng.standard_normal(K * R).astype(np.float32)
    ga, gb = hip.bern_lp_bwd(p, x, glp, K, R, D, logits=logits), orc.bern_lp_bwd(p, x, glp, K, R, D, logits=logits)
    close(ga["gp"], gb["gp"], 1e-4 if logits else 3e-5, 3e-5 * max(np.abs(gb["gp"]).max(), 1), "K3 bwd", shape)


def case_obs_grad():
    """zs_bernoulli_logprob_bwd_x (round 6): the gradient w.r.t. a differentiable observation of every period the product passes --
    one row per datapoint, full size, a single row, a scalar -- and periods that cut rows (the division path)."""
    K, R, D = shape3()
    logits, kfast = bool(rng.rand() < 0.4), bool(rng.rand() < 0.5)
    N = K * R * D
    p = (np.clip(2.0 * rng.standard_normal(N), -6, 6) if logits else rng.uniform(0.001, 0.999, N)).astype(np.float32)
    choice = rng.rand()
    if choice < 0.4:
        Px = R * D
    elif choice < 0.6:
        Px = N
    elif choice < 0.75:
        Px = D
    elif choice < 0.85:
        Px = 1
    else:                                   # any divisor of N
        divs = [d for d in range(1, min(N, 4096) + 1) if N % d == 0]
        Px = divs[int(rng.randint(len(divs)))]
    glp = rng.standard_normal(K * R).astype(np.float32)
    gscale = rng.standard_normal(R).astype(np.float32) if rng.rand() < 0.5 else None
    shape = (K, R, D, Px, logits, kfast, gscale is not None)
    a = hip.bern_lp_bwd_x(p, glp, K, R, D, Px, logits=logits, kfast=kfast, gscale=gscale)
    b = orc.bern_lp_bwd_x(p, glp, K, R, D, Px, logits=logits, kfast=kfast, gscale=gscale)
    # a sum of N / Px terms of size |g| * |log ratio| (up to ~14 for p at the clip): the fp32 sums differ by rounding order only
    close(a, b, 2e-4, 2e-5 * max(np.abs(b).max(), 1.0) + 3e-6 * (N // Px), "K3 bwd_x", shape)


def case_k4():
    """The gate of tests/test_cabi.py::_check_iw_reduce: the fp32 reference arithmetic (the oracle) loses accuracy as K grows, so the
    kernel is held to the FLOAT64 truth -- at least as close to it as 1.5 x the oracle -- and to the oracle within the oracle's
    own distance to the truth."""
    B, K = dim(3000), (dim(64) if rng.rand() < 0.7 else dim(1200))
    K = max(K, 2)
    spread = [1.0, 5.0, 30.0][int(rng.randint(3))]
    logp = (-550 + spread * rng.standard_normal((B, K))).astype(np.float32)
    logq = (-50 + 0.3 * spread * rng.standard_normal((B, K))).astype(np.float32)
    est = int(rng.randint(2))
    a, b, t = hip.iw(logp, logq, est), orc.iw(logp, logq, est), _iw_truth_f64(logp, logq, est)
    shape = (B, K, spread, est)
    close(a["bound"], t["bound"], 2e-6, 2e-5, "K4 bound", shape)
    close(a["cp"], t["cp"], 1e-4, 1e-6, "K4 cp", shape)
    if B * K < 2048 or B < 64:
        # a handful of elements or of rows: the maximum of either fp32 error is one unlucky rounding of (sum of the row - l) / (K - 1), a ~500-sized
        # intermediate of the reference's own formulation -- only a loose agreement is meaningful
        for key in ("cost", "cq"):
            close(a[key], b[key], 1e-4, 3e-6 * K * np.abs(logp - logq).max(), "K4 %s (small)" % key, shape)
    else:
        for key, floor in (("cost", 2e-5 * np.abs(t["cost"]).max()), ("cq", 2e-6)):
            err_hip, err_orc = np.abs(a[key] - t[key]).max(), np.abs(b[key] - t[key]).max()
            if not err_hip <= max(1.5 * err_orc, floor):
                print("MISMATCH K4 %s at %s: error against float64 %.3e, the fp32 oracle's %.3e" % (key, shape, err_hip, err_orc), flush=True)
                sys.exit(1)
            slack = 2.5 * err_orc + 1e-5 * max(1.0, np.abs(t[key]).max())
            if not np.abs(a[key] - b[key]).max() <= slack:
                print("MISMATCH K4 %s at %s: %.3e from the oracle (slack %.3e)" % (key, shape, np.abs(a[key] - b[key]).max(), slack), flush=True)
                sys.exit(1)
    close(hip.lme(logp), orc.lme(logp), 2e-6, 2e-5, "LME", shape)


def case_rng():
    n, s, off = dim(200000), int(rng.randint(1 << 30)), int(rng.randint(1 << 20))
    close(hip.philox(n, s, off), orc.philox(n, s, off), 0, 3e-5, "Philox normal", (n, s, off))
    P = dim(5000)
    p = rng.uniform(0, 1, P).astype(np.float32)
    nn = P * dim(20)
    if not np.array_equal(hip.bern_sample(p, nn, s, off), orc.bern_sample(p, nn, s, off)):
        print("MISMATCH K5 Bernoulli sample at %s" % ((P, nn, s, off),), flush=True)
        sys.exit(1)


def case_iw1():
    """IW1 over its whole domain (K 2..64, D a multiple of 4 in 256..1024, latent rows of whole 16-byte pieces): against the oracle, and --
    fed its own row sums -- bit for bit K4b; the merged backward against the oracle."""
    K, R, D = int(rng.randint(2, 65)), dim(300), 4 * int(rng.randint(64, 257))
    if rng.rand() < 0.3:                           # more datapoints than CUs: workgroups of the persistent kernel own 2 ... 12 of them
        R = int(rng.randint(257, 3100))
        while K > 2 and K * R * D > 3000000:
            K = max(K // 2, 2)
    while K * R * D > (3000000 if R > 256 else 1500000):
        R = max(R // 2, 1)
    Dz = 4 * int(rng.randint(1, 65 if rng.rand() < 0.2 else 12))
    logits, x_full, pms, pss, ls = [bool(rng.rand() < q) for q in (0.4, 0.3, 0.4, 0.4, 0.3)]
    with_z, with_rows = bool(rng.rand() < 0.8), bool(rng.rand() < 0.4)
    est, want_mean = int(rng.randint(2)), bool(rng.rand() < 0.7)
    f = lambda a: None if a is None else a.astype(np.float32)
    p, x, z, pmu, psg, rows_a, logq = iwf._inputs(rng, K, R, D, Dz, logits, x_full, pms, pss, ls)
    if rng.rand() < 0.3:                           # fractional observations in some rows (rows of bits take the one-logarithm form)
        frac = rng.uniform(size=x.shape)
        rows = rng.rand(*x.shape[:-1]) < 0.5
        x = np.where(rows[..., None], frac, x)
    args = (f(p), f(x), K, R, D, f(z) if with_z else None, f(pmu) if with_z else None, f(psg) if with_z else None, ls,
            f(rows_a) if with_rows else None, f(logq))
    shape = (K, R, D, Dz, logits, x_full, pms, pss, ls, with_z, with_rows, est, want_mean)
    got, ref = iwf.iw1(hip, *args, est, want_mean, logits), iwf.iw1(orc, *args, est, want_mean, logits)
    close(got["lp_x"], ref["lp_x"], 2e-5, 1e-3, "IW1 lp_x", shape)
    if with_z:
        close(got["lp_z"], ref["lp_z"], 2e-5, 1e-3, "IW1 lp_z", shape)
    close(got["bound"], ref["bound"], 2e-5, 1e-3, "IW1 bound", shape)
    same = iwf.composed(hip, *args, est, want_mean, logits, lp_x=got["lp_x"], lp_z=got["lp_z"])
    for key in ("cost", "bound", "coef"):
        if not np.array_equal(got[key], same[key]):
            print("MISMATCH IW1 %s differs from K4b on the same rows at %s" % (key, shape), flush=True)
            sys.exit(1)
    if want_mean:
        exact = got["cost"].astype(np.float64).mean()
        if not abs(got["mean"][0] - exact) <= 2e-6 * abs(exact):
            print("MISMATCH IW1 batch mean %r against %r at %s" % (got["mean"][0], exact, shape), flush=True)
            sys.exit(1)
    coef = (rng.standard_normal((2, R, K)) / R).astype(np.float32)
    gout = np.float32([rng.standard_normal()]) if rng.rand() < 0.5 else rng.standard_normal(R).astype(np.float32)
    q_ls = bool(rng.rand() < 0.3)
    qmu, qsg = f(rng.standard_normal((R, Dz)) * 0.3), f(rng.uniform(0.5, 1.5, size=(R, Dz)))
    if q_ls:
        qsg = np.log(qsg)
    a = iwf.iw1_bwd(hip, f(p), f(x), K, R, D, coef, gout, logits, f(z), qmu, qsg, q_ls)
    b = iwf.iw1_bwd(orc, f(p), f(x), K, R, D, coef, gout, logits, f(z), qmu, qsg, q_ls)
    close(a["gp"], b["gp"], 1e-4 if logits else 3e-5, 3e-6 * max(np.abs(b["gp"]).max(), 1e-30), "IW1 bwd gp", shape)
    for key in ("gqmu", "gqsigma"):
        close(a[key], b[key], 2e-4, 2e-5 * max(np.abs(b[key]).max(), 1e-30), "IW1 bwd " + key, shape)


cases = [case_k1, case_k2, case_k3, case_obs_grad, case_k4, case_rng, case_iw1, case_pair]
t0 = time.time()
while time.time() - t0 < budget:
    c = cases[int(rng.randint(len(cases)))]
    c()
    counts[c.__name__] = counts.get(c.__name__, 0) + 1
print("fuzz_hotpath: %.0f s, seed %d, no mismatch: %s" % (time.time() - t0, seed, counts))
