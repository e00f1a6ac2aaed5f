#!/usr/bin/env python
"""Size sweep of the hot-path kernels through the C ABI, timed with the library's kernel-bound HIP
events (zs_prof_*): algorithmic GB/s against the 8 TB/s HBM roofline.

  python tools/kernel_sweep.py [--out profiles/r01_kernel_sweep.json] [--max-rows 4194304]

Rows N = K*B with K = 50; D = 40 for the Normal kernels, X = 784 for the Bernoulli kernels.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))

import torch

from zhusuan import _hip

PEAK = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-rows", type=int, default=1 << 22)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--x-dim", type=int, default=784)
    ap.add_argument("--batches", type=int, nargs="*", default=None, help="batch sizes B (rows N = 50 B) instead of the default ladder")
    ap.add_argument("--only", action="append", default=None, help="only kernels whose name contains this substring (repeatable)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _hip.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = _hip.ptr
    K = 50
    results = []

    def timed(name, entry, bytes_, fn, note=""):
        if args.only and not any(o in name for o in args.only):
            return
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.prof_enable(True)
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        lib.prof_enable(False)
        q = lib.prof_query(entry)
        avg = q["total_ms"] / q["count"]
        each = sorted(1e3 * d for d in lib.prof_durations(entry))     # the same back-to-back launches, one by one
        med = each[len(each) // 2]
        rec = {"kernel": name, "entry": entry, "bytes": bytes_, "avg_us": 1e3 * avg, "min_us": 1e3 * q["min_ms"],
               "median_us": med, "launches": args.iters,
               "GBps": bytes_ / (avg * 1e-3) / 1e9, "frac_of_8TBps": bytes_ / (avg * 1e-3) / 1e9 / PEAK,
               "median_frac_of_8TBps": bytes_ / (med * 1e-6) / 1e9 / PEAK, "note": note}
        results.append(rec)
        print("%-34s %-26s %9.1f MB %9.2f us (min %8.2f, median %8.2f)  %7.1f GB/s  %5.1f%% (median %5.1f%%)" % (
            name, note, bytes_ / 1e6, rec["avg_us"], rec["min_us"], med, rec["GBps"], 100 * rec["frac_of_8TBps"],
            100 * rec["median_frac_of_8TBps"]), flush=True)

    Bs = args.batches or [256, 2621, 20971, 83886]          # N = K*B ~ 12800 (C3), 2^17, 2^20, 2^22
    for B in Bs:
        N = K * B
        if N > args.max_rows:
            continue
        # ---------------- K1 / K2: D = 40
        D = 40
        M = B * D
        mu = torch.randn(M, device=dev)
        sg = torch.rand(M, device=dev) + 0.5
        z = torch.empty(K * M, device=dev)
        lp = torch.empty(B * K, device=dev)
        eps = torch.randn(K * M, device=dev)
        tag = "N=%d (B=%d)" % (N, B)
        timed("K1 normal sample+lp (Philox)", "zs_normal_sample_logprob_f32", 4 * N * D + 4 * N + 8 * M,
              lambda: lib.call("zs_normal_sample_logprob_f32", P(mu), P(sg), None, 1, 2, None, P(z), P(lp), K, M, D, 1, K, 0, None, st), tag)
        timed("K1 normal sample+lp (eps given)", "zs_normal_sample_logprob_f32", 8 * N * D + 4 * N + 8 * M,
              lambda: lib.call("zs_normal_sample_logprob_f32", P(mu), P(sg), P(eps), 0, 0, None, P(z), P(lp), K, M, D, 1, K, 0, None, st), tag)
        timed("K2 normal logprob", "zs_normal_logprob_f32", 4 * N * D + 4 * N + 8 * M,
              lambda: lib.call("zs_normal_logprob_f32", P(z), K * M, P(mu), M, P(sg), M, P(lp), K, B, D, 1, K, 0, st), tag)
        gz = torch.randn(K * M, device=dev)
        glp = torch.randn(B * K, device=dev)
        gmu, gsg = torch.empty(M, device=dev), torch.empty(M, device=dev)
        timed("K1 bwd (reparam, Philox)", "zs_normal_sample_logprob_bwd_f32", 4 * N * D + 4 * N + 12 * M,
              lambda: lib.call("zs_normal_sample_logprob_bwd_f32", P(sg), None, 1, 2, None, P(gz), P(glp), 1, K, P(gmu), P(gsg), K, M, D, 0, st), tag)
        timed("K2 bwd ksum (non-reparam)", "zs_normal_logprob_bwd_ksum_f32", 4 * N * D + 4 * N + 16 * M,
              lambda: lib.call("zs_normal_logprob_bwd_ksum_f32", P(z), P(mu), P(sg), P(glp), 1, K, None, P(gmu), P(gsg), K, B, D, 0, st), tag)
        # ---------------- L1 / L2 / U1 / U2 (Logistic, Uniform): same shapes as K1 / K2
        u = eps.uniform_(1e-6, 1 - 1e-6)
        timed("L1 logistic sample+lp (Philox)", "zs_logistic_sample_logprob_f32", 4 * N * D + 4 * N + 8 * M,
              lambda: lib.call("zs_logistic_sample_logprob_f32", P(mu), P(sg), None, 1, 2, None, P(z), P(lp), K, M, D, 1, K, None, st), tag)
        timed("L1 logistic sample+lp (u given)", "zs_logistic_sample_logprob_f32", 8 * N * D + 4 * N + 8 * M,
              lambda: lib.call("zs_logistic_sample_logprob_f32", P(mu), P(sg), P(u), 0, 0, None, P(z), P(lp), K, M, D, 1, K, None, st), tag)
        timed("L2 logistic logprob", "zs_logistic_logprob_f32", 4 * N * D + 4 * N + 8 * M,
              lambda: lib.call("zs_logistic_logprob_f32", P(z), K * M, P(mu), M, P(sg), M, P(lp), K, B, D, 1, K, st), tag)
        timed("L1 bwd (Philox)", "zs_logistic_sample_logprob_bwd_f32", 4 * N * D + 4 * N + 12 * M,
              lambda: lib.call("zs_logistic_sample_logprob_bwd_f32", P(sg), None, 1, 2, None, P(gz), P(glp), 1, K, P(gmu), P(gsg), K, M, D, st), tag)
        timed("L2 bwd ksum (non-reparam)", "zs_logistic_logprob_bwd_ksum_f32", 4 * N * D + 4 * N + 16 * M,
              lambda: lib.call("zs_logistic_logprob_bwd_ksum_f32", P(z), P(mu), P(sg), P(glp), 1, K, None, P(gmu), P(gsg), K, B, D, st), tag)
        hi = mu + sg
        timed("U1 uniform sample (Philox)", "zs_uniform_sample_f32", 8 * N * D + 8 * M,
              lambda: lib.call("zs_uniform_sample_f32", P(mu), M, P(hi), M, None, 1, 2, None, P(z), P(gz), K * M, 1, st), tag)
        timed("U2 uniform logprob", "zs_uniform_logprob_f32", 4 * N * D + 4 * N + 8 * M,
              lambda: lib.call("zs_uniform_logprob_f32", P(z), K * M, P(mu), M, P(hi), M, P(lp), K, B, D, 1, K, st), tag)
        del eps, gz, z, u, hi
        # ---------------- K4
        logp = torch.randn(B, K, device=dev) - 550
        logq = torch.randn(B, K, device=dev) - 50
        cost, bound = torch.empty(B, device=dev), torch.empty(B, device=dev)
        cp, cq = torch.empty(B, K, device=dev), torch.empty(B, K, device=dev)
        timed("K4 iw reduce (vimco)", "zs_iw_reduce_f32", 16 * N + 8 * B,
              lambda: lib.call("zs_iw_reduce_f32", P(logp), K, P(logq), K, B, K, 1, P(cost), P(bound), P(cp), P(cq), st), tag)
        # ---------------- R1: the REINFORCE epilogue over N log-joints (variance reduction, batch mean): 8 N read, 4 N written
        mmt, stt = torch.zeros(1, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        sig, rcost = torch.empty(N, device=dev), torch.empty(1, device=dev)
        lpv, lqv = logp.reshape(-1), logq.reshape(-1)
        rws = torch.zeros(_hip.LJ_WORKSPACE, dtype=torch.float64, device=dev)
        rtk = torch.zeros(1, dtype=torch.int32, device=dev)
        timed("R1 reinforce epilogue", "zs_reinforce_f32", 12 * N,
              lambda: lib.call("zs_reinforce_f32", P(lpv), P(lqv), None, 1, N, 1, 1, 0.8, P(mmt), P(stt), P(sig), P(rcost), None,
                               P(rws), rws.numel(), P(rtk), st), tag + (" (2 launches: avg is per launch)" if N > 16384 else ""))
        # ---------------- A1: Adam (first size: the parameter tensors of the VAE / IWAE models; then one flat tensor)
        sizes = [392000, 500, 250000, 500, 20000, 40, 20000, 40, 20000, 500, 250000, 500, 392000, 784] if N == 12800 else [N * D]
        n_par = sum(sizes)
        a_p, a_g = [torch.randn(k, device=dev) for k in sizes], [torch.randn(k, device=dev) for k in sizes]
        a_m, a_v = torch.zeros(n_par, device=dev), torch.zeros(n_par, device=dev)
        a_step = torch.zeros(len(sizes), dtype=torch.int64, device=dev)
        a_ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        a_pp = (ctypes.c_void_p * len(sizes))(*[t.data_ptr() for t in a_p])
        a_gp = (ctypes.c_void_p * len(sizes))(*[t.data_ptr() for t in a_g])
        a_st = (ctypes.c_int64 * (len(sizes) + 1))(*([sum(sizes[:i]) for i in range(len(sizes))] + [n_par]))
        timed("A1 adam step", "zs_adam_step_f32", 28 * n_par,
              lambda: lib.call("zs_adam_step_f32", a_pp, a_gp, a_st, len(sizes), P(a_m), P(a_v), P(a_step), P(a_ticket), n_par, 1e-3,
                               0.9, 0.999, 1e-8, 1.0, None, st), "n=%d in %d tensors" % (n_par, len(sizes)))
        del a_p, a_g, a_m, a_v
        # ---------------- K3: X = 784 (--x-dim: experiments on the row length)
        X = args.x_dim
        if N * X * 4 * 2 > 200e9:
            continue
        p = torch.rand(N * X, device=dev) * 0.96 + 0.02
        x = (torch.rand(B * X, device=dev) < 0.5).float()
        timed("K3 bernoulli logprob", "zs_bernoulli_logprob_f32", 4 * N * X + 4 * B * X + 4 * N,
              lambda: lib.call("zs_bernoulli_logprob_f32", P(p), P(x), B * X, P(lp), K, B, X, 1, K, st), tag)
        timed("K3 bernoulli logprob (logits)", "zs_bernoulli_logits_logprob_f32", 4 * N * X + 4 * B * X + 4 * N,
              lambda: lib.call("zs_bernoulli_logits_logprob_f32", P(p), P(x), B * X, P(lp), None, K, B, X, 1, K, st), tag)
        gp = torch.empty(N * X, device=dev)
        timed("K3 bwd", "zs_bernoulli_logprob_bwd_f32", 8 * N * X + 4 * B * X + 4 * N,
              lambda: lib.call("zs_bernoulli_logprob_bwd_f32", P(p), P(x), B * X, P(glp), 1, K, P(gp), K, B, X, st), tag)
        timed("K3 bwd (logits)", "zs_bernoulli_logits_logprob_bwd_f32", 8 * N * X + 4 * B * X + 4 * N,
              lambda: lib.call("zs_bernoulli_logits_logprob_bwd_f32", P(p), P(x), B * X, P(glp), 1, K, P(gp), K, B, X, st), tag)
        gx = torch.empty(B * X, device=dev)
        timed("K3 bwd_x (observation gradient)", "zs_bernoulli_logprob_bwd_x_f32", 4 * N * X + 4 * N + 4 * B * X,
              lambda: lib.call("zs_bernoulli_logprob_bwd_x_f32", P(p), 0, B * X, P(glp), 1, K, None, 0, P(gx), K, B, X, st), tag)
        del p, gp, x, gx
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0), "peak_GBps": PEAK, "K": K, "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
