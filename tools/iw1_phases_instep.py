#!/usr/bin/env python
"""tools/iw1_phases.py's stamps read after GRAPH REPLAYS of the whole C3 training step (experiments build): where the persistent
IW1 forward kernel's time goes when it runs where it is used -- behind the decoder's kernels, cold instruction cache, the stream
freshly written -- against the same kernel launched back to back in the same process.

  ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so python tools/iw1_phases_instep.py [batch=256]
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
import zhusuan          # noqa: E402
from zhusuan import _hip      # noqa: E402
from examples import iwae     # noqa: E402

NAMES = ["start", "w0 rows requested", "first row landed (w0)", "prologue done", "w0@last barrier", "barrier passed", "reduction done", "share out"]
ORDER = [0, 7, 6, 1, 2, 3, 4, 5]


def read_stamps(klib, G):
    read = klib.cdll.zs_iw1_stamps_read
    read.restype = ctypes.c_int
    read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    buf = np.zeros(G * 8, dtype=np.uint64)
    assert read(buf.ctypes.data, G * 8) == 0
    s = buf.reshape(G, 8).astype(np.int64)
    rw = klib.cdll.zs_iw1_wave_stamps_read
    rw.restype = ctypes.c_int
    rw.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    wb = np.zeros(G * 32, dtype=np.uint64)
    assert rw(wb.ctypes.data, G * 32) == 0
    return s, wb.reshape(G, 32).astype(np.int64)


def report(label, samples, K):
    a = np.median(np.stack([(s[:, ORDER] - s[:, 0].min()) * 0.01 for s, _ in samples]), axis=0)
    ends = [((s[:, 5].max() - s[:, 0].min()) * 0.01) for s, _ in samples]
    print("%s: us from the earliest workgroup's start (median over %d launches of each workgroup's offset)" % (label, len(samples)))
    for j, nm in enumerate(NAMES):
        c = a[:, j]
        print("  %-22s median %7.2f   min %7.2f   max %7.2f" % (nm, np.median(c), c.min(), c.max()))
    print("  kernel end (latest share out): median %.2f  min %.2f  max %.2f" % (np.median(ends), min(ends), max(ends)))
    nw = min(K, 16)
    fr = np.median(np.stack([(w[:, 16:16 + nw] - s[:, 0].min()) * 0.01 for s, w in samples]), axis=(0, 1))
    lb = np.median(np.stack([(w[:, :nw] - s[:, 0].min()) * 0.01 for s, w in samples]), axis=(0, 1))
    print("  per wave: first row landed | arrival at the last barrier")
    print("   " + " ".join("w%-2d %5.2f|%5.2f" % (i, fr[i], lb[i]) for i in range(nw)))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = 50
    dev = torch.device("cuda", 0)
    klib = _hip.lib()
    assert hasattr(klib.cdll, "zs_iw1_stamps_read"), "experiments build needed: ZS_HIP_LIBRARY=tools/_exp/libzs_hip_exp.so"
    G = min(B, torch.cuda.get_device_properties(dev).multi_processor_count)
    torch.manual_seed(0)
    model, obs = iwae.build(K, "vimco", device=dev, dense="fused"), {"x": (torch.rand(B, 784, device=dev) < 0.5).float()}
    opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    rng = zhusuan.DeviceRNG(dev, seed=1)

    def compute():
        rng.begin_step()
        for p in model.parameters():
            p.grad = None
        loss = model(obs)
        loss.backward()
        return loss.detach()
    step = zhusuan.GraphedStep(compute, opt.step, rng=rng)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    samples = []
    for _ in range(25):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        samples.append(read_stamps(klib, G))
    report("B=%d K=%d inside the graph-replayed training step" % (B, K), samples, K)
    # the same kernel back to back (tools/iw1_phases.py's setting) in this process
    P = _hip.ptr
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    X, D, N = 784, 40, K * B
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
    samples = []
    for _ in range(25):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        samples.append(read_stamps(klib, G))
    report("the same shape launched back to back", samples, K)
    # ... and behind a producer: an elementwise pass that rewrites the stream right before each launch (what the decoder's sigmoid does)
    logits = torch.randn(N * X, device=dev)
    samples = []
    for _ in range(25):
        for _ in range(3):
            torch.sigmoid(logits, out=p)
            fn()
        torch.cuda.synchronize()
        samples.append(read_stamps(klib, G))
    report("the same shape, each launch behind torch.sigmoid(out=p)", samples, K)


if __name__ == "__main__":
    main()
