#!/usr/bin/env python
"""Randomised differential run of the callers' layer kernels (CS1, AB1, PR1, PL1, PM1) and the multi-node sampler's backward (MS1,
gz + gz2) through the C ABI: libzs_hip.so on the GPU against the C oracle on the host, random shapes -- odd column counts, single
rows, tiles that end ragged, widths at the LDS limit -- for a given number of seconds.  Uses the raw-call helpers of
tests/test_logjoint.py.  Exit code 1 at the first mismatch (the failing shape is printed).

  python tools/fuzz_layers.py [seconds=120] [seed=0]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

from zhusuan import _hip
from zhusuan.layers import _fits_lds, _fits_lds_mlp
from conftest import host_kernel_library
from test_logjoint import Raw

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


def rand_dim(hi, small=0.3):
    if rng.rand() < small:
        return int(rng.randint(1, min(hi, 9) + 1))
    return int(rng.randint(1, hi + 1))


def case_colsum():
    rows, cols = (rand_dim(20000) if rng.rand() < 0.5 else rand_dim(300)), rand_dim(1100)
    if rng.rand() < 0.05:
        rows = 0
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    a, b = hip.colsum(x), orc.colsum(x)
    close(a, b, 2e-6, 4e-6 * max(np.sqrt(rows), 1), "CS1", (rows, cols))
    if rows:
        act = int(rng.randint(1, 3))
        pre = rng.standard_normal((rows, cols)).astype(np.float32)
        y = np.maximum(pre, 0) if act == 1 else (1.0 / (1.0 + np.exp(-2.0 * pre))).astype(np.float32)
        inp = bool(rng.rand() < 0.3)
        (ga, sa), (gb, sb) = hip.actbwd(x, y, act, in_place=inp), orc.actbwd(x, y, act)
        close(ga, gb, 3e-7, 1e-30, "AB1 gpre", (rows, cols, act))
        close(sa, sb, 2e-6, 4e-6 * max(np.sqrt(rows), 1), "AB1 bias", (rows, cols, act))


def case_rmse():
    K, B = rand_dim(64), (rand_dim(20000) if rng.rand() < 0.5 else rand_dim(600))
    pred, y = rng.standard_normal((K, B)).astype(np.float32), rng.standard_normal(B).astype(np.float32)
    # (the particle mean is summed in a different order: its rounding shows where y - mean cancels)
    close(hip.rmse(pred, y), orc.rmse(pred, y), 3e-6, 3e-7 * (np.abs(y).max() + np.abs(pred).max()), "PR1", (K, B))


def case_pl():
    while True:
        n_in, n_out = rand_dim(255), rand_dim(256)
        if _fits_lds(n_in, n_out, 4):
            break
    K, B = rand_dim(12), rand_dim(400)
    shared, relu = bool(rng.rand() < 0.5), bool(rng.rand() < 0.5)
    h = rng.standard_normal((B, n_in) if shared else (K, B, n_in)).astype(np.float32)
    w = (rng.standard_normal((K, n_out, n_in + 1)) / np.sqrt(n_in + 1)).astype(np.float32)
    gout = rng.standard_normal((K, B, n_out)).astype(np.float32)
    oa, ob = hip.pl(h, w, relu), orc.pl(h, w, relu)
    close(oa, ob, 3e-5, 3e-5, "PL1 fwd", (K, B, n_in, n_out, shared, relu))
    want_gh = bool(rng.rand() < 0.6)
    (gha, gwa), (ghb, gwb) = hip.pl_bwd(h, w, ob, gout, relu, want_gh), orc.pl_bwd(h, w, ob, gout, relu, want_gh)
    close(gwa, gwb, 3e-4, 3e-4 * max(np.abs(gwb).max(), 1), "PL1 gw", (K, B, n_in, n_out, shared, relu))
    if want_gh:
        close(gha, ghb, 3e-4, 3e-4 * max(np.abs(ghb).max(), 1), "PL1 gh", (K, B, n_in, n_out, shared, relu))


def case_pm():
    while True:
        L = int(rng.randint(1, 5))
        sizes = tuple(rand_dim(96, small=0.4) for _ in range(L + 1))
        if _fits_lds_mlp(sizes, 4):
            break
    K, B = rand_dim(12), rand_dim(300)
    shared = bool(rng.rand() < 0.5)
    x = rng.standard_normal((B, sizes[0]) if shared else (K, B, sizes[0])).astype(np.float32)
    ws = [(rng.standard_normal((K, sizes[l + 1], sizes[l] + 1)) / np.sqrt(sizes[l] + 1)).astype(np.float32) for l in range(L)]
    gout = rng.standard_normal((K, B, sizes[-1])).astype(np.float32)
    oa, ob = hip.pm(x, ws), orc.pm(x, ws)
    for l in range(L):
        close(oa[l], ob[l], 5e-5, 5e-5, "PM1 out%d" % l, (sizes, K, B, shared))
    # the device chain of PL1 calls: bit-identical
    h = x
    for l, w in enumerate(ws):
        h = hip.pl(h, w, l < L - 1)
        if not np.array_equal(h, oa[l]):
            print("MISMATCH PM1 != PL1 chain (layer %d) at %s" % (l, (sizes, K, B, shared)), flush=True)
            sys.exit(1)
    want_gx = bool(rng.rand() < 0.5)
    (gxa, gwa), (gxb, gwb) = hip.pm_bwd(x, ws, oa, gout, want_gx), orc.pm_bwd(x, ws, oa, gout, want_gx)
    for l in range(L):
        close(gwa[l], gwb[l], 5e-4, 5e-4 * max(np.abs(gwb[l]).max(), 1), "PM1 gw%d" % l, (sizes, K, B, shared))
    if want_gx:
        close(gxa, gxb, 5e-4, 5e-4 * max(np.abs(gxb).max(), 1), "PM1 gx", (sizes, K, B, shared))


def case_ms_bwd():
    nodes = []
    for _ in range(int(rng.randint(1, 6))):
        K, R, D, ls = rand_dim(20), rand_dim(12), rand_dim(300), int(rng.rand() < 0.5)
        M = R * D
        nd = {"mu": rng.standard_normal(M), "sigma": rng.uniform(-0.5, 0.3, M) if ls else rng.uniform(0.5, 1.5, M), "K": K, "D": D,
              "ls": ls, "offset": 7 * len(nodes), "kfast": bool(rng.rand() < 0.5), "eps": rng.standard_normal(K * M)}
        mode = int(rng.randint(0, 3))                                    # gz only / gz2 only / both
        if mode != 1:
            nd["gz"] = rng.standard_normal(K * M)
        if mode != 0:
            nd["gz2"] = rng.standard_normal(K * M)
        if rng.rand() < 0.7:
            nd["glp"] = rng.standard_normal(K * R)
        nodes.append(nd)
    st_h, st_o = torch.tensor([5, 77], dtype=torch.int64, device=hip.dev), torch.tensor([5, 77], dtype=torch.int64)
    for (ga, sa), (gb, sb), nd in zip(hip.ms_bwd(nodes, rs=st_h), orc.ms_bwd(nodes, rs=st_o), nodes):
        shape = (nd["K"], nd["mu"].size, nd["D"], nd["ls"], "gz" in nd, "gz2" in nd, "glp" in nd)
        close(ga, gb, 2e-5, 2e-5 * max(np.abs(gb).max(), 1), "MS1 gmu", shape)
        close(sa, sb, 3e-4, 3e-4 * max(np.abs(sb).max(), 1), "MS1 gsigma", shape)


cases = [case_colsum, case_rmse, case_pl, case_pm, case_ms_bwd]
t0 = time.time()
while time.time() - t0 < budget:
    c = cases[int(rng.randint(len(cases)))]
    c()
    counts[c.__name__] = counts.get(c.__name__, 0) + 1
print("fuzz_layers: %.0f s, seed %d, no mismatch: %s" % (time.time() - t0, seed, counts))
