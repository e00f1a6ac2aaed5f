"""Is a hipGraph replay of the training step THE SAME COMPUTATION as its eager launches -- on every one of N consecutive steps, not at a few
checkpoints?  The IWAE step of bench.py from the same weights, optimizer state and Philox state, once replayed from a graph and once
launched from Python; the objective of EVERY step is kept and the two sequences are compared bit for bit (a race between kernels of a
replay -- a missing edge, a stale cache line -- would show up as a first step where they part).  For FlatAdam and for
torch.optim.Adam(fused, capturable).  Also counted: spikes of the objective (a step more than 1.5 x the median of the 200 before it).

    python tools/graph_vs_eager_long.py [steps]          -> profiles/r06_graph_vs_eager_long.txt
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    model0, obs, _, _ = bench.make_workload("c3", dev)
    one = torch.ones((), device=dev)

    def run(kind, graphed):
        model = copy.deepcopy(model0)
        opt = bench.make_optimizer(model, kind == "torch.Adam")
        rng = zhusuan.DeviceRNG(dev, seed=1)
        params = list(model.parameters())

        def compute():
            rng.begin_step()
            for p in params:
                p.grad = None
            loss = model(obs)
            loss.backward(one)
            return loss.detach()
        out = torch.zeros(steps, device=dev)
        with zhusuan.device_rng(rng):
            if graphed:
                step = zhusuan.GraphedStep(compute, opt.step, rng=rng, warmup=3, restore=True, optimizer=opt)
                for i in range(steps):
                    out[i] = step()
            else:
                if kind == "torch.Adam":       # (the graphed twin's warm-up created torch's state tensors and restore zeroed them: same start)
                    pass
                for i in range(steps):
                    out[i] = compute()
                    opt.step()
        torch.cuda.synchronize()
        return out.cpu(), [p.detach().clone() for p in params]

    def spikes(x):
        n = 0
        onsets = []
        i = 200
        while i < len(x):
            med = float(x[i - 200:i].median())
            if float(x[i]) > 1.5 * med:
                n += 1
                onsets.append(i)
                i += 500            # one event
            else:
                i += 1
        return n, onsets

    print("c3 (K = 50, B = 256), lr 1e-3, %d steps from the same weights / optimizer state / Philox state" % steps)
    for kind in ("FlatAdam", "torch.Adam"):
        g, pg = run(kind, True)
        e, pe = run(kind, False)
        same = (g == e)
        n_diff = int((~same).sum())
        first = int((~same).nonzero()[0]) if n_diff else -1
        print("%-10s graph vs eager: %d of %d objectives differ%s; parameters at the end equal: %s" % (
            kind, n_diff, steps, "" if n_diff == 0 else " (first at step %d: %.6f vs %.6f)" % (first, float(g[first]), float(e[first])),
            all(torch.equal(a, b) for a, b in zip(pg, pe))))
        for name, x in (("graph", g), ("eager", e)):
            n, on = spikes(x)
            print("    %s: mean objective of the last 1000 steps %.3f; spikes: %d%s" % (name, float(x[-1000:].mean()), n,
                  "" if not on else "  at steps %s" % on[:12]))
            for o in on[:2]:
                print("        around step %d: %s" % (o, " ".join("%.1f" % float(v) for v in x[max(o - 6, 0):o + 10])))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
