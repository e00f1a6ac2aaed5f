#!/usr/bin/env python
"""Where a workgroup of the persistent IW1 forward kernel spends its time (experiments build: ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so).
In-kernel s_memrealtime stamps (100 MHz) per workgroup -> median / min / max offsets from the EARLIEST kernel start, in us:
  start, wave 0's rows requested, its observation row landed, prologue done, wave 0 at the last datapoint's barrier, barrier passed, K-particle reduction done, share out.

  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so python tools/iw1_phases.py [B K] ...
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np      # noqa: E402
import torch            # noqa: E402
from zhusuan import _hip      # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    klib = _hip.lib()
    P = _hip.ptr
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    read = klib.cdll.zs_iw1_stamps_read
    read.restype = ctypes.c_int
    read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(256, 50), (512, 50), (1024, 50), (2048, 10)]
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    names = ["start", "w0 rows requested", "w0 x landed", "prologue done", "w0@last barrier", "barrier passed", "reduction done", "share out"]
    order = [0, 7, 6, 1, 2, 3, 4, 5]
    for B, K in shapes:
        X, D = 784, 40
        N = K * B
        p = torch.rand(N * X, device=dev) * 0.96 + 0.02
        x = (torch.rand(B * X, device=dev) < 0.5).float()
        z = torch.randn(N * D, device=dev)
        mu, sg = torch.zeros(B * D, device=dev), torch.ones(B * D, device=dev)
        logq = torch.randn(B * K, device=dev) - 45
        lpx, lpz = torch.empty(B * K, device=dev), torch.empty(B * K, device=dev)
        cost, bound, coef = torch.empty(1, device=dev), torch.empty(B, device=dev), torch.empty(2 * B * K, device=dev)
        costb, tk = torch.empty(B, device=dev), torch.zeros(64, dtype=torch.int64, device=dev)
        fn = lambda: klib.call("zs_bernoulli_iw_objective_f32", P(p), 0, P(x), B * X, K, B, X, P(z), P(mu), B * D, P(sg), B * D, D, 0, None, K,
                               P(logq), K, 1, 1, P(lpx), P(lpz), P(costb), P(bound), P(coef), P(cost), P(tk), st)
        G = min(B, n_cu)
        rows = []
        for rep in range(12):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            buf = np.zeros(G * 8, dtype=np.uint64)
            assert read(buf.ctypes.data, G * 8) == 0
            s = buf.reshape(G, 8).astype(np.int64)
            t0 = s[:, 0].min()
            rows.append((s[:, order] - t0) * 0.01)          # us
        a = np.median(np.stack(rows), axis=0)                 # per workgroup: median over the repetitions
        print("B=%d K=%d  (%d workgroups, %d datapoints each at most); us from the earliest start" % (B, K, G, -(-B // G)))
        for j, nm in enumerate(names):
            c = a[:, j]
            print("  %-18s median %7.2f   min %7.2f   max %7.2f" % (nm, np.median(c), c.min(), c.max()))
        print("  kernel end (latest share out) %.2f" % a[:, 7].max())
        if hasattr(klib.cdll, "zs_iw1_wave_stamps_read"):
            rw = klib.cdll.zs_iw1_wave_stamps_read
            rw.restype = ctypes.c_int
            rw.argtypes = [ctypes.c_void_p, ctypes.c_int64]
            wb = np.zeros(G * 32, dtype=np.uint64)
            assert rw(wb.ctypes.data, G * 32) == 0
            ws = wb.reshape(G, 32).astype(np.int64)
            buf0 = np.zeros(G * 8, dtype=np.uint64)
            read(buf0.ctypes.data, G * 8)
            t00 = buf0.reshape(G, 8).astype(np.int64)[:, 0].min()
            nw = min(K, 16)
            print("  per wave (last launch), median over workgroups, us:  first row landed | arrival at the last barrier")
            print("   " + " ".join("w%-2d %5.2f|%5.2f" % (i, np.median((ws[:, 16 + i] - t00) * 0.01), np.median((ws[:, i] - t00) * 0.01)) for i in range(nw)))


if __name__ == "__main__":
    main()
