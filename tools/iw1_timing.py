#!/usr/bin/env python
"""IW1 (zs_bernoulli_iw_objective: generator side of the IW objective in one launch) against the three launches it replaces
(K3 forward + K2 + K4b), and its backward call against K3 backward + K2 backward-ksum, through the C ABI with HIP events bound
to each dispatch (median of `--launches` back-to-back launches).  Shapes: the config sizes (C3: B=256 K=50; the reference
example's default B=64 K=40) and larger batches up to the fused kernel's row limit.

  python tools/iw1_timing.py [--launches 50] [--cold]     (--cold: a 512 MB fill between launches, so that inputs come from HBM)
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch      # noqa: E402
from zhusuan import _hip      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=50)
    ap.add_argument("--cold", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    klib = _hip.lib()
    P = _hip.ptr
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    big = torch.empty(128 << 20, device=dev) if a.cold else None

    def timed(entries, fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        klib.prof_enable(True)
        for _ in range(a.launches):
            if big is not None:
                big.fill_(1.0)
            fn()
        torch.cuda.synchronize()
        klib.prof_enable(False)
        out = []
        for e in entries:
            d = sorted(1e3 * v for v in klib.prof_durations(e))
            out.append(d[len(d) // 2] if d else float("nan"))
        return out

    print("%-22s %9s %9s %9s %9s | %9s || %9s %9s %9s | %9s" % ("shape", "K3 fwd", "K2", "K4b", "sum", "IW1", "K3 bwd", "ksum", "sum", "IW1 bwd"))
    for B, K in ((64, 40), (256, 50), (384, 50), (512, 50), (1024, 50), (2048, 50), (2048, 10)):
        X, D = 784, 40
        N = K * B
        p = torch.rand(N * X, device=dev) * 0.96 + 0.02
        x = (torch.rand(B * X, device=dev) < 0.5).float()
        z = torch.randn(N * D, device=dev)
        mu, sg = torch.zeros(B * D, device=dev), torch.ones(B * D, device=dev)
        qmu, qsg = torch.randn(B * D, device=dev), torch.rand(B * D, device=dev) + 0.5
        logq = torch.randn(B * K, device=dev) - 45
        lpx, lpz = torch.empty(B * K, device=dev), torch.empty(B * K, device=dev)
        cost, bound, coef = torch.empty(1, device=dev), torch.empty(B, device=dev), torch.empty(2 * B * K, device=dev)
        ws, tk = torch.empty(max(B, 4096), device=dev), torch.zeros(64, dtype=torch.int64, device=dev)
        costb = torch.empty(B, device=dev)
        tk1 = torch.zeros(1, dtype=torch.int32, device=dev)
        gp = torch.empty(N * X, device=dev)
        gm, gs = torch.empty(B * D, device=dev), torch.empty(B * D, device=dev)
        g = torch.ones(1, device=dev)
        f3, = timed(["zs_bernoulli_logprob_f32"], lambda: klib.call("zs_bernoulli_logprob_f32", P(p), P(x), B * X, P(lpx), K, B, X, 1, K, st))
        f2, = timed(["zs_normal_logprob_f32"], lambda: klib.call("zs_normal_logprob_f32", P(z), N * D, P(mu), B * D, P(sg), B * D, P(lpz), K, B, D, 1, K, 0, st))
        f4, = timed(["zs_iw_objective_f32"], lambda: klib.call("zs_iw_objective_f32", P(lpz), K, P(lpx), K, P(logq), K, B, K, 1, 1, None, P(bound), P(coef), P(cost),
                                                                P(ws), ws.numel(), P(tk1), st))
        f1, = timed(["zs_bernoulli_iw_objective_f32"], lambda: klib.call(
            "zs_bernoulli_iw_objective_f32", P(p), 0, P(x), B * X, K, B, X, P(z), P(mu), B * D, P(sg), B * D, D, 0, None, K, P(logq), K, 1, 1,
            P(lpx), P(lpz), P(costb), P(bound), P(coef), P(cost), P(tk), st))
        assert int(tk.abs().sum()) == 0
        b3, = timed(["zs_bernoulli_logprob_bwd_f32"], lambda: klib.call("zs_bernoulli_logprob_bwd_f32", P(p), P(x), B * X, P(coef), 1, K, P(gp), K, B, X, st))
        bk, = timed(["zs_normal_logprob_bwd_ksum_f32"], lambda: klib.call("zs_normal_logprob_bwd_ksum_f32", P(z), P(qmu), P(qsg), P(coef[B * K:]), 1, K, None,
                                                                             P(gm), P(gs), K, B, D, 0, st))
        ik, = timed(["zs_bernoulli_iw_objective_bwd_f32"], lambda: klib.call(
            "zs_bernoulli_iw_objective_bwd_f32", P(p), 0, P(x), B * X, K, B, X, P(coef), P(g), 0, P(gp), P(z), P(qmu), P(qsg), D, 0, P(gm), P(gs), st))
        print("B=%-5d K=%-3d %7.0f MB %9.2f %9.2f %9.2f %9.2f | %9.2f || %9.2f %9.2f %9.2f | %9.2f" % (
            B, K, 4e-6 * N * X, f3, f2, f4, f3 + f2 + f4, f1, b3, bk, b3 + bk, ik), flush=True)
        del p, gp
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
