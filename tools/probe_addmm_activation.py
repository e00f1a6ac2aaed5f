#!/usr/bin/env python
"""Which kernels torch._addmm_activation (bias + ReLU epilogue) launches on this build, next to addmm + relu: does the dense
layer's ReLU ride in the GEMM's epilogue here?   python tools/probe_addmm_activation.py [--tuned]"""
import sys
import torch
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
tuned = "--tuned" in sys.argv
if tuned:
    import tempfile, os
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_filename(os.path.join(tempfile.mkdtemp(), "t.csv"))
for rows, n_in, n_out in ((12800, 500, 500), (12800, 40, 500), (256, 784, 500), (512, 500, 500)):
    x = torch.randn(rows, n_in, device=dev)
    w = torch.randn(n_out, n_in, device=dev) * 0.05
    b = torch.randn(n_out, device=dev)
    variants = {
        "linear+relu": lambda: torch.relu(torch.nn.functional.linear(x, w, b)),
        "_addmm_activation": lambda: torch._addmm_activation(b, x, w.t()),
    }
    ref = variants["linear+relu"]()
    for name, fn in variants.items():
        for _ in range(5):
            out = fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        acc = {}
        for e in prof.events():
            d = getattr(e, "device_time", 0.0)
            if d and d > 0:
                a = acc.setdefault(e.name, [0, 0.0]); a[0] += 1; a[1] += d
        print("[%d,%d]x[%d]  %-18s max|diff| %.2e  %s" % (rows, n_in, n_out, name, float((out - ref).abs().max()),
              "; ".join("%.0fx %.1f us %s" % (n / 20, t / n, k[:60]) for k, (n, t) in acc.items())), flush=True)
