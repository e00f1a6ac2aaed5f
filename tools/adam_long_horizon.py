"""Long horizons on ONE resident minibatch: the IWAE step of bench.py (single hipGraph) for `steps` updates with zhusuan.optim.FlatAdam
and with torch.optim.Adam(fused, capturable), from the same initial weights, for several Philox seeds; every `every` steps the mean
objective of the last 200 steps, the largest |parameter| and the smallest / largest second moment.  (profiles/r06_dp_soak.txt: after
180 000 steps FlatAdam's run stood at 579, torch's at 60 -- one sample each.  Is that the optimizer or the dynamics?)

    python tools/adam_long_horizon.py [steps] [every] [seeds] [first seed]          -> profiles/r06_adam_long_horizon.txt
"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                              # noqa: E402
import torch                              # noqa: E402


def main():
    import zhusuan
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    first_seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    model0, obs, _, _ = bench.make_workload("c3", dev)
    one = torch.ones((), device=dev)
    print("c3 (K = 50, B = 256), lr 1e-3, one resident minibatch; mean objective of the last 200 steps | max |param| | min .. max of exp_avg_sq")
    for seed in range(first_seed, first_seed + seeds):
        for kind in ("FlatAdam", "torch.Adam"):
            model = copy.deepcopy(model0)
            opt = bench.make_optimizer(model, kind == "torch.Adam")
            rng = zhusuan.DeviceRNG(dev, seed=seed)

            def compute():
                rng.begin_step()
                for p in model.parameters():
                    p.grad = None
                loss = model(obs)
                loss.backward(one)
                return loss.detach()
            with zhusuan.device_rng(rng):
                step = zhusuan.GraphedStep(compute, opt.step, rng=rng, warmup=3)
            acc = torch.zeros((), device=dev, dtype=torch.float64)
            every_loss = torch.zeros(steps, device=dev)
            row = []
            for i in range(steps):
                loss = step()
                every_loss[i] = loss
                if (i % every) >= every - 200:
                    acc += loss
                if (i + 1) % every == 0:
                    pmax = max(float(p.detach().abs().max()) for p in model.parameters())
                    if kind == "FlatAdam":
                        v = torch.cat([b.exp_avg_sq for b in opt.buckets])
                    else:
                        v = torch.cat([s["exp_avg_sq"].reshape(-1) for s in opt.state.values()])
                    row.append("%9.3f |%7.2f |%8.1e ..%8.1e" % (float(acc) / 200, pmax, float(v.min()), float(v.max())))
                    acc.zero_()
            x, n_spikes, onsets, i = every_loss.cpu(), 0, [], 200
            while i < steps:          # a spike: one step's objective more than 1.5 x the median of the 200 before it (one event per 500 steps)
                if float(x[i]) > 1.5 * float(x[i - 200:i].median()):
                    n_spikes += 1
                    onsets.append(i)
                    i += 500
                else:
                    i += 1
            print("seed %d  %-10s spikes of the objective: %d%s" % (seed, kind, n_spikes, ("  at steps %s" % onsets[:16]) if onsets else ""))
            for j, r in enumerate(row):
                print("    %7d  %s" % ((j + 1) * every, r))
            sys.stdout.flush()
            del step


if __name__ == "__main__":
    main()
