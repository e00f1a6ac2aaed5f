#!/usr/bin/env python
"""ELBO-evals/s of the other BASELINE.json configs on one MI355X, with the CPU oracle beside them:
C2 VAE-MNIST SGVB B=512 K=1, C3 IWAE VIMCO / SGVB B=256 K=50 (also with the sigmoid fused into the
Bernoulli kernel), C5 (per GPU) BNN-VI B=512 K=10.  Full training steps (fwd + bwd + Adam), hipGraph replay, with
bench.py's settings: the callers' dense layers are zhusuan.Linear in zhusuan.Sequential, the draw the reference's objectives discard is skipped
(zhusuan.skip_discarded_draws), TunableOp picks the callers' fp32 GEMM solutions.  --reference-settings: torch.nn.Linear,
the discarded draw executed, PyTorch's default GEMM selection (what round 2's table measured).

  python tools/bench_configs.py [--steps 200] [--out profiles/r01_configs.json] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch

import zhusuan
import bench as B
from examples import vae_mnist, iwae, bnn_vi

REFERENCE_SETTINGS = False


def run_gpu(name, build, make_obs, evals_per_step, steps, graph=True, forward_only=False):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tuned = B.gemm_tuning(not REFERENCE_SETTINGS)
    model = build(dev)
    opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    obs = make_obs(dev)
    rng = zhusuan.DeviceRNG(dev, seed=1)
    one = torch.ones((), device=dev)

    def body():
        rng.begin_step()
        if forward_only:                      # objective evaluation only (SURVEY 8d metric (i))
            with torch.no_grad():
                return model(obs).detach()
        for p in model.parameters():
            p.grad = None
        loss = model(obs)
        loss.backward(one)
        opt.step()
        return loss.detach()

    with zhusuan.device_rng(rng), zhusuan.skip_discarded_draws(not REFERENCE_SETTINGS):
        step = body
        if not graph:
            for _ in range(10):
                body()
        else:
            def compute():
                rng.begin_step()
                if forward_only:
                    with torch.no_grad():
                        return model(obs).detach()
                for p in model.parameters():
                    p.grad = None
                loss = model(obs)
                loss.backward(one)
                return loss.detach()
            step = zhusuan.GraphedStep(compute, None if forward_only else opt.step, rng=rng, warmup=10)
        B.gemm_tuning(tuned, tune=False)     # every GEMM shape of the step has been seen: keep the picks, stop timing
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rec = {"config": name, "device": "MI355X", "ms_per_step": 1e3 * dt / steps, "evals_per_s": evals_per_step * steps / dt,
           "launch_mode": "hipgraph" if graph else "eager", "final_loss": float(last),
           "settings": "reference (torch.nn.Linear, discarded draws executed, default GEMM selection)" if REFERENCE_SETTINGS else
                       "bench.py's (zhusuan.Linear + zhusuan.Sequential, discarded draws skipped, TunableOp)"}
    print(json.dumps(rec), flush=True)
    return rec


def run_cpu(name, step_fn, evals_per_step, budget=8.0, max_steps=50):
    avail = len(os.sched_getaffinity(0))
    best = None
    for nt in sorted({c for c in (8, 16, 32, 64) if c <= avail}):
        torch.set_num_threads(nt)
        step_fn()
        t0 = time.perf_counter()
        step_fn()
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
    torch.set_num_threads(best[0])
    n, t0 = 0, time.perf_counter()
    while n < max_steps and time.perf_counter() - t0 < budget:
        step_fn()
        n += 1
    dt = time.perf_counter() - t0
    rec = {"config": name, "device": "host CPU, torch-CPU oracle (oracle/zs_oracle.py)", "threads": best[0],
           "ms_per_step": 1e3 * dt / n, "evals_per_s": evals_per_step * n / dt, "steps": n}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--reference-settings", action="store_true")
    ap.add_argument("--only", default=None, help="run only the GPU configs whose name contains this substring")
    args = ap.parse_args()
    global REFERENCE_SETTINGS
    REFERENCE_SETTINGS = args.reference_settings
    dense = "torch" if REFERENCE_SETTINGS else "fused"
    res = []
    real_run_gpu = run_gpu

    def run_gpu_filtered(name, *a, **k):
        if args.only and args.only not in name:
            return {"config": name, "skipped": True}
        return real_run_gpu(name, *a, **k)
    globals()["run_gpu"] = run_gpu_filtered
    bits = lambda B: (lambda dev: {"x": (torch.rand(B, 784, device=dev) < 0.5).float()})
    res.append(run_gpu("C2 VAE SGVB B=512 K=1", lambda d: vae_mnist.build(512, device=d, dense=dense), bits(512), 512, args.steps))
    res.append(run_gpu("C1-shape VAE SGVB B=64 K=1", lambda d: vae_mnist.build(64, device=d, dense=dense), bits(64), 64, args.steps))
    res.append(run_gpu("C3 IWAE VIMCO B=256 K=50", lambda d: iwae.build(50, "vimco", device=d, dense=dense), bits(256), 12800, args.steps))
    res.append(run_gpu("C3 IWAE VIMCO B=256 K=50 (eager launches)", lambda d: iwae.build(50, "vimco", device=d, dense=dense), bits(256), 12800,
                       args.steps, graph=False))
    res.append(run_gpu("C3 IWAE VIMCO B=256 K=50, Bernoulli from logits (sigmoid fused)",
                       lambda d: iwae.build(50, "vimco", device=d, fused_logits=True, dense=dense), bits(256), 12800, args.steps))
    res.append(run_gpu("C3 IWAE VIMCO B=256 K=50, objective forward only", lambda d: iwae.build(50, "vimco", device=d, dense=dense), bits(256),
                       12800, args.steps, forward_only=True))
    res.append(run_gpu("C2 VAE SGVB B=512 K=1, objective forward only", lambda d: vae_mnist.build(512, device=d, dense=dense), bits(512), 512,
                       args.steps, forward_only=True))
    res.append(run_gpu("C3 IWAE SGVB B=256 K=50", lambda d: iwae.build(50, "sgvb", device=d, dense=dense), bits(256), 12800, args.steps))
    bnn_obs = lambda B: (lambda dev: {"x": torch.randn(B, 13, device=dev), "y": torch.randn(B, device=dev)})
    res.append(run_gpu("C5/GPU BNN-VI B=512 K=10", lambda d: bnn_vi.build(n_particles=10, device=d), bnn_obs(512), 5120, args.steps))
    res.append(run_gpu("C5 global BNN-VI B=4096 K=10", lambda d: bnn_vi.build(n_particles=10, device=d), bnn_obs(4096), 40960, args.steps))
    res.append(run_gpu("reference default BNN-VI B=114 K=512", lambda d: bnn_vi.build(n_particles=512, device=d), bnn_obs(114),
                       114 * 512, args.steps))
    if not args.no_cpu:
        from oracle import zs_oracle as O
        import helpers as H
        # VAE B=512
        spec = H.vae_param_spec()
        p = H.make_params(spec, 1)
        opt = torch.optim.Adam([p[n] for n, _ in spec], 1e-3)
        x = (torch.rand(512, 784) < 0.5).float()

        def vae_step():
            torch.randn(512, 40)
            loss, _ = O.vae_loss(p, x, torch.randn(512, 40))
            opt.zero_grad(); loss.backward(); opt.step()
        res.append(run_cpu("C2 VAE SGVB B=512 K=1", vae_step, 512))
        # BNN B=512 K=10
        wm, wl, yl = H.bnn_params(512, 10)
        opt2 = torch.optim.Adam(wm + wl + [yl], 1e-3)
        xb, yb = torch.randn(512, 13), torch.randn(512)

        def bnn_step():
            eps = [torch.randn(10, 50, 14), torch.randn(10, 1, 51)]
            torch.randn(10, 50, 14); torch.randn(10, 1, 51)
            loss, _ = O.bnn_loss(wm, wl, yl, xb, yb, eps, 10)
            opt2.zero_grad(); loss.backward(); opt2.step()
        res.append(run_cpu("C5/GPU BNN-VI B=512 K=10", bnn_step, 5120))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
