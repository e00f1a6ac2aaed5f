#!/usr/bin/env python
"""Soak of the three example steps: 20 000 hipGraph replays + 300 eager steps each (bench.py's settings: fused dense layers,
discarded draws skipped), then the invariants of the hand-off kernels -- every ticket word of every scratch set and of the
optimizer back at zero, Adam's per-tensor step counts equal to the number of steps, finite loss, no allocator growth."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch

import zhusuan
from zhusuan import _ops
from examples import iwae, vae_mnist, bnn_vi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")
ok = True
for kind in ("iwae", "vae", "bnn"):
    torch.manual_seed(0)
    if kind == "iwae":
        model, obs = iwae.build(50, "vimco", device=dev, dense="fused"), {"x": (torch.rand(256, 784, device=dev) < 0.5).float()}
    elif kind == "vae":
        model, obs = vae_mnist.build(512, device=dev, dense="fused"), {"x": (torch.rand(512, 784, device=dev) < 0.5).float()}
    else:
        model, obs = bnn_vi.build(n_particles=10, device=dev), {"x": torch.randn(512, 13, device=dev), "y": torch.randn(512, device=dev)}
    opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    rng = zhusuan.DeviceRNG(dev, seed=1)

    def compute():
        rng.begin_step()
        for p in model.parameters():
            p.grad = None
        loss = model(obs)
        loss.backward()
        return loss.detach()
    with zhusuan.skip_discarded_draws():
        step = zhusuan.GraphedStep(compute, opt.step, rng=rng)
        l0 = float(step())
        m0 = torch.cuda.memory_allocated()
        steps0 = [b.step.clone() for b in opt.buckets]
        t0 = time.perf_counter()
        for i in range(N):
            last = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        with zhusuan.device_rng(rng):          # eager steps too (allocator churn, the per-stream scratch sets)
            for i in range(300):
                compute()
                opt.step()
    torch.cuda.synchronize()
    counted = all(bool(((b.step - s0) == N + 300).all()) for b, s0 in zip(opt.buckets, steps0))
    tickets = [int(b.ticket.item()) for b in opt.buckets]
    scratch = 0
    for key, tensors in _ops._SCRATCH.items():
        for t in tensors:
            if t.dtype == torch.int32:
                scratch += int(t.abs().sum().item())
    good = counted and not any(tickets) and scratch == 0 and bool(torch.isfinite(last))
    ok = ok and good
    print("%-4s %s  loss %.2f -> %.2f  %.4f ms/step  Adam step counts advanced by %d: %s  optimizer tickets %s  scratch tickets (sum) %d  "
          "mem delta %d B" % (kind, "OK  " if good else "FAIL", l0, float(last), 1e3 * dt / N, N + 300, counted, tickets, scratch,
                              torch.cuda.memory_allocated() - m0), flush=True)
sys.exit(0 if ok else 1)
