#!/usr/bin/env python
"""Durations of the one-launch kernels (LJ1 / MS1 / PL1) at the BNN (config 5 per GPU) and VAE (config 2) shapes, launched
back to back through the C ABI with the library's own per-dispatch HIP events (zs_prof_*): median / min of 50 launches.
For kernel experiments:  ZS_HIP_LIBRARY=/path/to/variant.so python tools/small_kernels_timing.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

from zhusuan import _hip
from test_logjoint import Raw, _term, LJ


def main():
    dev = torch.device("cuda:0")
    lib = _hip.lib()
    raw = Raw(lib, dev)
    st = raw.stream()
    P = _hip.ptr
    rng = np.random.RandomState(0)
    print("library:", lib.path, "|", lib.build_info())

    def timed(label, entry, fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        lib.prof_enable(True)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        lib.prof_enable(False)
        d = sorted(1e3 * v for v in lib.prof_durations(entry))
        print("%-58s median %6.2f us   min %6.2f us" % (label, d[len(d) // 2], d[0]))

    # ---- PL1 at the two layers of config 5 (K = 10, B = 512)
    for (K, B, n_in, n_out, shared, relu) in [(10, 512, 13, 50, True, True), (10, 512, 50, 1, False, False),
                                              (10, 4096, 13, 50, True, True), (10, 4096, 50, 1, False, False)]:
        h = torch.randn((B, n_in) if shared else (K, B, n_in), device=dev)
        w = torch.randn(K, n_out, n_in + 1, device=dev)
        out = torch.empty(K, B, n_out, device=dev)
        gout = torch.randn(K, B, n_out, device=dev)
        gh = torch.empty(K, B, n_in, device=dev)
        gw = torch.empty_like(w)
        part = torch.empty(K * ((B + 15) // 16) * n_out * (n_in + 1), device=dev)
        tk = torch.zeros(64, dtype=torch.int32, device=dev)
        hsk = 0 if shared else B * n_in
        tag = "K=%d B=%d %d->%d" % (K, B, n_in, n_out)
        timed("PL1 fwd " + tag, "zs_particle_linear_f32",
              lambda: lib.call("zs_particle_linear_f32", P(h), hsk, P(w), P(out), K, B, n_in, n_out, int(relu), st))
        timed("PL1 bwd " + tag + (" (gw only)" if shared else ""), "zs_particle_linear_bwd_f32",
              lambda: lib.call("zs_particle_linear_bwd_f32", P(h), hsk, P(w), P(out), P(gout), None if shared else P(gh), P(gw), K, B,
                               n_in, n_out, int(relu), P(part), part.numel(), P(tk), st))
    # ---- PM1: config 5's whole network [13, 50, 1] in one launch each way
    for (K, B) in [(10, 512), (10, 4096)]:
        sizes = (13, 50, 1)
        x = torch.randn(B, sizes[0], device=dev)
        ws = [torch.randn(K, sizes[l + 1], sizes[l] + 1, device=dev) for l in range(2)]
        outs = [torch.empty(K, B, sizes[l + 1], device=dev) for l in range(2)]
        gws = [torch.empty_like(w) for w in ws]
        gout = torch.randn(K, B, 1, device=dev)
        slab = sum(w.shape[1] * w.shape[2] + 3 for w in ws)
        part = torch.empty(K * ((B + 15) // 16) * slab, device=dev)
        tk = torch.zeros(64, dtype=torch.int32, device=dev)
        table = (_hip.PMLayer * 2)()
        for l in range(2):
            table[l].w, table[l].out, table[l].gw = P(ws[l]), P(outs[l]), P(gws[l])
            table[l].n_in, table[l].n_out = sizes[l], sizes[l + 1]
        tag = "K=%d B=%d 13->50->1" % (K, B)
        timed("PM1 fwd " + tag, "zs_particle_mlp_f32",
              lambda: lib.call("zs_particle_mlp_f32", P(x), 0, ctypes.byref(table), 2, K, B, st))
        timed("PM1 bwd " + tag, "zs_particle_mlp_bwd_f32",
              lambda: lib.call("zs_particle_mlp_bwd_f32", P(x), 0, ctypes.byref(table), 2, P(gout), None, K, B, P(part), part.numel(), P(tk), st))
    # ---- CS1: the bias gradients of the IWAE (12 800 rows) and VAE (512 rows) steps
    for rows, cols in [(12800, 500), (12800, 784), (12800, 40), (512, 500), (512, 784)]:
        x = torch.randn(rows, cols, device=dev)
        o = torch.empty(cols, device=dev)
        ws = torch.empty(128 * (cols + 256), device=dev)
        tk = torch.zeros(16, dtype=torch.int32, device=dev)
        timed("CS1 column sum [%d, %d]  (%.1f MB)" % (rows, cols, rows * cols * 4 / 1e6), "zs_column_sum_f32",
              lambda: lib.call("zs_column_sum_f32", P(x), P(o), rows, cols, P(ws), ws.numel(), P(tk), tk.numel(), st))
        yv, gp = torch.relu(torch.randn(rows, cols, device=dev)), torch.empty(rows, cols, device=dev)
        timed("AB1 relu backward + column sum [%d, %d]" % (rows, cols), "zs_dense_act_bwd_f32",
              lambda: lib.call("zs_dense_act_bwd_f32", P(x), P(yv), 1, P(gp), P(o), rows, cols, P(ws), ws.numel(), P(tk), tk.numel(), st))
        timed("AB1 sigmoid backward + column sum [%d, %d]" % (rows, cols), "zs_dense_act_bwd_f32",
              lambda: lib.call("zs_dense_act_bwd_f32", P(x), P(yv), 2, P(gp), P(o), rows, cols, P(ws), ws.numel(), P(tk), tk.numel(), st))
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            x.sum(0)
        t0.record()
        for _ in range(50):
            x.sum(0)
        t1.record(); torch.cuda.synchronize()
        print("%-58s %6.2f us per call (stream time, 50 calls back to back)" % ("   torch x.sum(0) of the same matrix", t0.elapsed_time(t1) * 1e3 / 50))
    # ---- LJ1: the BNN objective's five terms and the VAE objective's three
    N, NL, Bn, R = LJ.LJ_NORMAL, LJ.LJ_NORMAL_LOGSTD, LJ.LJ_BERNOULLI, LJ.LJ_ROWS
    cases = {
        "BNN (7000 + 510 + 5120 elements, 2 x 10 rows)": [
            _term(rng, N, 10 * 700, (None, 700, 700), coef=-0.1, want=(True, False, False)),
            _term(rng, N, 10 * 51, (None, 51, 51), coef=-0.1, want=(True, False, False)),
            _term(rng, NL, 10 * 512, (512, None, 1), coef=-456. / 5120, want=(False, True, True)),
            _term(rng, R, 10, coef=0.1), _term(rng, R, 10, coef=0.1)],
        "VAE B=512 (20480 + 401408 elements, 512 rows)": [
            _term(rng, N, 512 * 40, coef=-1 / 512., want=(True, False, False)),
            _term(rng, Bn, 512 * 784, coef=-1 / 512., want=(False, True, False)), _term(rng, R, 512, coef=1 / 512.)],
    }
    for label, terms in cases.items():
        grads = {}
        tab, keep = raw._table(terms, grads)
        out = torch.empty(1, device=dev)
        ws = torch.zeros(_hip.LJ_WORKSPACE, dtype=torch.float64, device=dev)
        tk = torch.zeros(1, dtype=torch.int32, device=dev)
        g = torch.ones(1, device=dev)
        gc = torch.empty(len(terms), device=dev)
        timed("LJ1 fwd " + label, "zs_logjoint_scalar_f32",
              lambda: lib.call("zs_logjoint_scalar_f32", ctypes.byref(tab), len(terms), P(out), P(ws), ws.numel(), P(tk), st))
        timed("LJ1 bwd " + label, "zs_logjoint_scalar_bwd_f32",
              lambda: lib.call("zs_logjoint_scalar_bwd_f32", ctypes.byref(tab), len(terms), P(g), P(gc), P(ws), ws.numel(), P(tk), st))
    # ---- MS1: the BNN's two weight matrices
    tab = (_hip.MSTerm * 2)()
    keep = []
    for i, (K, M, D) in enumerate([(10, 700, 700), (10, 51, 51)]):
        mu, sg = torch.randn(M, device=dev), torch.randn(M, device=dev) * 0.1
        z, lp = torch.empty(K, M, device=dev), torch.empty(K, device=dev)
        gz, glp = torch.randn(K, M, device=dev), torch.randn(K, device=dev)
        gmu, gs = torch.empty(M, device=dev), torch.empty(M, device=dev)
        e = tab[i]
        e.mu, e.sigma, e.z, e.lp = mu.data_ptr(), sg.data_ptr(), z.data_ptr(), lp.data_ptr()
        e.K, e.M, e.D, e.lp_stride_k, e.lp_stride_r, e.offset, e.sigma_is_logstd = K, M, D, 1, 1, i, 1
        e.gz, e.glp, e.glp_stride_k, e.glp_stride_r, e.gmu, e.gsigma = gz.data_ptr(), glp.data_ptr(), 1, 1, gmu.data_ptr(), gs.data_ptr()
        keep += [mu, sg, z, lp, gz, glp, gmu, gs]
    timed("MS1 fwd BNN w0 [10,50,14] + w1 [10,1,51]", "zs_normal_sample_logprob_multi_f32",
          lambda: lib.call("zs_normal_sample_logprob_multi_f32", ctypes.byref(tab), 2, 1, None, None, st))
    timed("MS1 bwd", "zs_normal_sample_logprob_multi_bwd_f32",
          lambda: lib.call("zs_normal_sample_logprob_multi_bwd_f32", ctypes.byref(tab), 2, 1, None, st))


if __name__ == "__main__":
    main()
