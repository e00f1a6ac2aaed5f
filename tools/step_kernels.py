#!/usr/bin/env python
"""Per-step kernel table of one config: which kernels one training step launches, how often and for how long, split
into hot-path kernels of this package (k_*) and everything else (the callers' GEMMs, torch element-wise glue, Adam).

  python tools/step_kernels.py --config C5 [--steps 100] [--out profiles/r02_step_kernels_C5.json]
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_C5 -- python3 tools/step_kernels.py --config C5 --external

Runs exactly `--steps` eager steps (no hipGraph: a replay hides nothing from a tracer, but eager keeps the launch order
readable).  In-process the device timestamps come from torch.profiler (roctracer); with --external the process only
runs the steps and an outside tracer (rocprofv3) does the counting: Calls / steps = launches per step.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch

import zhusuan
from examples import vae_mnist, iwae, bnn_vi


def make(config, dev, bnn_layer="fused", dense="fused"):
    bits = lambda B: {"x": (torch.rand(B, 784, device=dev) < 0.5).float()}
    if config == "C2":
        return vae_mnist.build(512, device=dev, dense=dense), bits(512), 512
    if config == "C3":
        return iwae.build(50, "vimco", device=dev, dense=dense), bits(256), 12800
    if config == "C3-logits":
        return iwae.build(50, "vimco", device=dev, fused_logits=True, dense=dense), bits(256), 12800
    if config == "C5":
        return bnn_vi.build(n_particles=10, device=dev, layer=bnn_layer), {"x": torch.randn(512, 13, device=dev), "y": torch.randn(512, device=dev)}, 5120
    raise SystemExit("unknown config %s" % config)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5", choices=["C2", "C3", "C3-logits", "C5"])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--external", action="store_true", help="an outside tracer counts; just run the steps")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused, capturable) instead of zhusuan.optim.FlatAdam")
    ap.add_argument("--tuned-gemm", action="store_true", help="PyTorch TunableOp picks the callers' GEMM solutions, as bench.py does")
    ap.add_argument("--skip-discarded-draws", action="store_true",
                    help="run inside zhusuan.skip_discarded_draws() (an opt-in; default: both draws executed, as the package and bench.py do)")
    ap.add_argument("--reference-draws", action="store_true", help="(the default since round 4; kept for old command lines)")
    ap.add_argument("--bnn-layer", default="fused", choices=["fused", "per_layer", "bmm", "materialize"])
    ap.add_argument("--dense", default="fused", choices=["fused", "zhusuan", "torch"],
                    help="fused: zhusuan.Linear in zhusuan.Sequential (AB1); zhusuan: zhusuan.Linear (CS1 bias gradient) in nn.Sequential; torch: torch.nn")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, obs, evals = make(args.config, dev, args.bnn_layer, args.dense)
    if args.tuned_gemm:
        import tempfile
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_filename(os.path.join(tempfile.mkdtemp(), "tunableop.csv"))
    opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True, capturable=True) if args.torch_adam else \
        zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    rng = zhusuan.DeviceRNG(dev, seed=1)

    one = torch.ones((), device=dev)          # backward's seed, allocated once (as bench.py does)

    def step():
        rng.begin_step()
        for p in model.parameters():
            p.grad = None
        loss = model(obs)
        loss.backward(one)
        opt.step()

    with zhusuan.device_rng(rng), zhusuan.skip_discarded_draws(args.skip_discarded_draws):
        if args.external:
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            return
        # the step replayed from a hipGraph, timed from the host (what bench.py's extra_configs report).  FIRST: autograd's
        # AccumulateGrad nodes remember the stream of their first use, and eager steps on the default stream before a capture
        # break it (zhusuan.GraphedStep warms up on its own capture stream)
        import time

        def compute():
            rng.begin_step()
            for p in model.parameters():
                p.grad = None
            loss = model(obs)
            loss.backward(one)
            return loss.detach()
        gstep = zhusuan.GraphedStep(compute, opt.step, rng=rng, warmup=5)
        if args.tuned_gemm:
            torch.cuda.tunable.tuning_enable(False)
        for _ in range(20):
            gstep()
        torch.cuda.synchronize()
        reps = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200):
                gstep()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 200)
        graph_ms = 1e3 * sorted(reps)[len(reps) // 2]
        side = torch.cuda.Stream()           # eager launches next to a captured graph: stay off the default stream
        side.wait_stream(torch.cuda.current_stream())
        from torch.profiler import profile, ProfilerActivity
        with torch.cuda.stream(side):
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
    acc = {}
    for e in prof.events():
        d = getattr(e, "device_time", None) or getattr(e, "cuda_time", 0.0)
        if d and d > 0:
            a = acc.setdefault(e.name, [0, 0.0])
            a[0] += 1
            a[1] += float(d)
    rows = []
    for name, (n, tot) in acc.items():
        ours = "k_" in name and ("zs" in name or name.lstrip("void ").startswith(("(anonymous namespace)::k_", "k_")))
        # kernels of this package: HOT = the distribution / objective / update kernels of the hot path; LAYER = the BNN caller's
        # layers (PL1, CS1 / AB1, PR1: the caller's code in the reference, kernels of this package here)
        kind = "" if not ours else ("LAYER" if any(f in name for f in ("k_particle_linear", "k_particle_mlp", "k_column_sum", "k_particle_rmse")) else "HOT")
        rows.append({"kernel": name[:140], "hot_path": kind == "HOT", "kind": kind or "other", "launches_per_step": n / args.steps,
                     "avg_us": tot / n, "us_per_step": tot / args.steps})
    rows.sort(key=lambda r: -r["us_per_step"])

    def tot(kind):
        sel = [r for r in rows if r["kind"] == kind]
        return {"launches_per_step": sum(r["launches_per_step"] for r in sel), "us_per_step": sum(r["us_per_step"] for r in sel)}
    summary = {"config": args.config, "steps": args.steps, "launch_mode": "eager (per-kernel rows); hipgraph_ms_per_step: replayed",
               "discarded_draws": "skipped (zhusuan.skip_discarded_draws)" if args.skip_discarded_draws else "executed (package default)",
               "hipgraph_ms_per_step": graph_ms,
               "hot_path": tot("HOT"), "caller_layer_kernels_of_this_package": tot("LAYER"), "other": tot("other"), "kernels": rows}
    print("%s (%s): hot-path %.1f launches / %.1f us per step; this package's layer kernels %.1f / %.1f us; other (torch, BLAS) %.1f launches "
          "/ %.1f us per step; the step replayed from a hipGraph: %.4f ms" % (
              args.config, "discarded draws skipped" if args.skip_discarded_draws else "both draws executed",
              summary["hot_path"]["launches_per_step"], summary["hot_path"]["us_per_step"],
              summary["caller_layer_kernels_of_this_package"]["launches_per_step"],
              summary["caller_layer_kernels_of_this_package"]["us_per_step"],
              summary["other"]["launches_per_step"], summary["other"]["us_per_step"], graph_ms))
    for r in rows:
        print("  %-5s %5.2f x %8.2f us  %s" % (r["kind"].upper() if r["kind"] != "other" else "", r["launches_per_step"], r["avg_us"],
                                            r["kernel"][:110]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
