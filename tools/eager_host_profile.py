#!/usr/bin/env python
"""Where the HOST time of an eagerly launched training step goes.

A step launched from Python (no hipGraph) is bound by the launch path, not by the GPU: this tool times the eager step of a
BASELINE config in several settings, and ranks the host side with cProfile.

  python tools/eager_host_profile.py [--config c3|c5|c2] [--steps 300] [--profile]

Settings (one line each):
  reference example   torch.nn modules, torch.optim.Adam (its defaults), default GEMM selection, both draws, obs refreshed
  bench settings      zhusuan.Linear/Sequential, FlatAdam, DeviceRNG, discarded draws executed
  ... + skip          the same inside zhusuan.skip_discarded_draws()
`--profile` adds a cProfile table (top entries by own time) of the first setting.
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np      # noqa: E402
import torch            # noqa: E402


def build(config, dev, dense):
    from examples import iwae, vae_mnist, bnn_vi
    rs = np.random.RandomState(1234)
    bits = lambda B: torch.tensor((rs.uniform(size=(B, 784)) < 0.5).astype(np.float32), device=dev)
    if config == "c3":
        return iwae.build(n_samples=50, estimator="vimco", device=dev, dense=dense), {"x": bits(256)}, 256 * 50
    if config == "c3d":       # the reference example's own defaults (iwae.py:126,131)
        return iwae.build(n_samples=40, estimator="vimco", device=dev, dense=dense), {"x": bits(64)}, 64 * 40
    if config == "c2":
        return vae_mnist.build(512, device=dev, dense=dense), {"x": bits(512)}, 512
    if config == "c5":
        x = torch.tensor(rs.standard_normal((512, 13)).astype(np.float32), device=dev)
        y = torch.tensor(rs.standard_normal(512).astype(np.float32), device=dev)
        return bnn_vi.build(n_particles=10, device=dev), {"x": x, "y": y}, 5120
    raise ValueError(config)


def make_step(config, dev, dense, adam, device_rng, skip, refresh):
    import zhusuan
    torch.manual_seed(0)
    model, obs, evals = build(config, dev, dense)
    if adam == "torch":
        opt = torch.optim.Adam(model.parameters(), 1e-3)
    elif adam == "torch_fused":
        opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True)
    else:
        opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    rng = zhusuan.DeviceRNG(dev, seed=1) if device_rng else None
    fresh = dict((k, v.clone()) for k, v in obs.items())

    def step():
        if refresh:
            for k in obs:
                obs[k].copy_(fresh[k], non_blocking=True)
        if rng is not None:
            rng.begin_step()
        loss = model(obs)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    import contextlib
    ctx = contextlib.ExitStack()
    if rng is not None:
        ctx.enter_context(zhusuan.device_rng(rng))
    if skip:
        ctx.enter_context(zhusuan.skip_discarded_draws())
    return step, ctx, evals


def time_steps(step, n):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = time.perf_counter() - t0          # until Python has issued everything
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return 1e3 * t_host / n, 1e3 * t_all / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--null-kernels", action="store_true", help="also time with this package's kernel launches skipped (host cost only)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    settings = [
        ("reference example (torch.nn, torch Adam, both draws, obs refreshed)", dict(dense="torch", adam="torch", device_rng=False, skip=False, refresh=True)),
        ("  + torch Adam fused=True", dict(dense="torch", adam="torch_fused", device_rng=False, skip=False, refresh=True)),
        ("  + FlatAdam", dict(dense="torch", adam="flat", device_rng=False, skip=False, refresh=True)),
        ("bench settings eager (fused dense, FlatAdam, DeviceRNG, both draws)", dict(dense="fused", adam="flat", device_rng=True, skip=False, refresh=False)),
        ("  + skip_discarded_draws", dict(dense="fused", adam="flat", device_rng=True, skip=True, refresh=False)),
    ]
    if args.config == "c5":
        settings = [s for s in settings if s[1]["dense"] == "torch" or "bench" in s[0] or "skip" in s[0]]
    for label, kw in settings:
        step, ctx, evals = make_step(args.config, dev, **kw)
        with ctx:
            host, total = time_steps(step, args.steps)
            print("%-78s host %.3f ms  step %.3f ms" % (label, host, total), flush=True)
            if args.null_kernels:
                from zhusuan import _hip
                klib = _hip.lib()
                real = klib.call
                klib.call = lambda name, *a: None
                try:
                    host0, total0 = time_steps(step, args.steps)
                finally:
                    klib.call = real
                print("%-78s host %.3f ms  step %.3f ms   (package launches skipped)" % ("", host0, total0), flush=True)
            if args.profile and label.startswith(("reference", "bench")):
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(args.steps):
                    step()
                pr.disable()
                torch.cuda.synchronize()
                s = io.StringIO()
                pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
                print(s.getvalue())
        del step
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
