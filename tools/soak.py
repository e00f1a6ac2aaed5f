#!/usr/bin/env python
"""Soak of the three example steps (the package's default settings: both draws of every latent executed): N (20 000) hipGraph
replays + 300 eager steps each.  What it ASSERTS (exit code 1 otherwise):

  values      every N / 10 replays the replayed step is compared with an EAGER evaluation of the same step on a twin model
              (same parameters, same Philox state): the loss and every gradient tensor -- the cross-workgroup hand-offs
              (LJ1 / MS1 / PM1 / AB1 / CS1 partials + tickets, IW1's fixed-point batch mean, FlatAdam's step counter) reduce
              the SAME numbers after thousands of launches on one workspace as a fresh eager launch on another;
  hand-offs   every ticket / accumulator word of every scratch set and of the optimizer back at zero; Adam's per-tensor step
              counts advanced by exactly the number of steps;
  allocator   torch.cuda.memory_allocated() does not move across any window of N / 10 replays (measured from the end of one twin
              evaluation to the start of the next: a replay allocates nothing; the twin's own eager step is kept out of the window --
              its fresh gradient tensors land in cached blocks of slightly different sizes, +-0.7 MB from one evaluation to the
              next), and the eager steps return to their own baseline once the last loss is dropped (round 3's "160 MB delta" of
              the IWAE line was the live autograd graph of the last eager step: the tool measured before dropping it);
  finite loss.

  python tools/soak.py [N]
"""
import copy
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import numpy as np
import torch

import zhusuan
from zhusuan import _ops
from examples import iwae, vae_mnist, bnn_vi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")
ok = True


def settled():
    gc.collect()
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated()


for kind in ("iwae", "vae", "bnn"):
    torch.manual_seed(0)
    if kind == "iwae":
        model, obs = iwae.build(50, "vimco", device=dev, dense="fused"), {"x": (torch.rand(256, 784, device=dev) < 0.5).float()}
    elif kind == "vae":
        model, obs = vae_mnist.build(512, device=dev, dense="fused"), {"x": (torch.rand(512, 784, device=dev) < 0.5).float()}
    else:
        model, obs = bnn_vi.build(n_particles=10, device=dev), {"x": torch.randn(512, 13, device=dev), "y": torch.randn(512, device=dev)}
    twin = copy.deepcopy(model)
    opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    rng, rng_t = zhusuan.DeviceRNG(dev, seed=1), zhusuan.DeviceRNG(dev, seed=1)

    def make_compute(m, r):
        def compute():
            r.begin_step()
            for p in m.parameters():
                p.grad = None
            loss = m(obs)
            loss.backward()
            return loss.detach()
        return compute
    compute, compute_t = make_compute(model, rng), make_compute(twin, rng_t)
    step = zhusuan.GraphedStep(compute, opt.step, rng=rng)
    l0 = float(step())
    steps0 = [b.step.clone() for b in opt.buckets]
    worst, checks, window_start, replay_growth, last = 0.0, 0, None, 0, None
    t0 = time.perf_counter()
    for i in range(N):
        check = i % max(N // 10, 1) == 0
        if check:                       # the twin takes the parameters and the RNG state the replay is about to use
            with torch.no_grad():
                for pt, pm in zip(twin.parameters(), model.parameters()):
                    pt.copy_(pm)
                rng_t.state.copy_(rng.state)
        last = step()
        if check:
            if window_start is not None:
                replay_growth = max(replay_growth, abs(settled() - window_start))
            with zhusuan.device_rng(rng_t):
                lt = compute_t()
            torch.cuda.synchronize()
            err = abs(float(last) - float(lt)) / max(abs(float(lt)), 1e-30)
            for pm, pt in zip(model.parameters(), twin.parameters()):
                a, b = pm.grad.double(), pt.grad.double()
                err = max(err, float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)))
            worst, checks = max(worst, err), checks + 1
            del lt, a, b
            window_start = settled()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    replay_growth = max(replay_growth, abs(settled() - window_start))
    with zhusuan.device_rng(rng):          # eager steps too (allocator churn, the per-stream scratch sets)
        loss_e = compute()
        opt.step()
        del loss_e
        for p in model.parameters():
            p.grad = None
        mem_e0 = settled()
        for i in range(299):
            loss_e = compute()
            opt.step()
        del loss_e
        for p in model.parameters():
            p.grad = None
        mem_e1 = settled()
    counted = all(bool(((b.step - s0) == N + 300).all()) for b, s0 in zip(opt.buckets, steps0))
    tickets = [int(b.ticket.item()) for b in opt.buckets]
    scratch = 0
    for key, tensors in list(_ops._SCRATCH.items()) + [(None, t) for t in _ops._SCRATCH_RETIRED]:
        for t in tensors:
            if t.dtype in (torch.int32, torch.int64):
                scratch += int(t.abs().sum().item())
    values_ok = worst <= 2e-4          # (eager and replayed kernels are the same code: differences are summation order of atomics-free sums)
    good = counted and not any(tickets) and scratch == 0 and bool(torch.isfinite(last)) and values_ok and replay_growth == 0 and mem_e1 <= mem_e0
    ok = ok and good
    print("%-4s %s  loss %.2f -> %.2f  %.4f ms/step  Adam step counts advanced by %d: %s  optimizer tickets %s  scratch tickets (sum) %d  "
          "replay == eager twin at %d checkpoints: worst rel. diff %.2e  allocator: largest change over a window of %d replays: %d B, eager step 1 -> 300: %+d B" % (
              kind, "OK  " if good else "FAIL", l0, float(last), 1e3 * dt / N, N + 300, counted, tickets, scratch, checks, worst, max(N // 10, 1),
              replay_growth, mem_e1 - mem_e0), flush=True)
sys.exit(0 if ok else 1)
