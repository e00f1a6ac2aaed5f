"""FlatAdam against torch.optim.Adam(fused, capturable) ON THE SAME GRADIENTS: the IWAE step of bench.py trains model A with FlatAdam; every
step A's gradients are copied to a twin B (same initial weights) that torch's Adam updates -- B's parameters never feed back into a
gradient, so whatever separates A from B is the optimizers' arithmetic alone, not the training dynamics.  Printed every `every` steps:
the largest |difference| of parameters / first / second moments relative to the largest magnitude, and the number of parameters that
are more than one learning-rate step (1e-3) apart.

    python tools/adam_side_by_side.py [steps] [every] [eager|graph]          -> profiles/r06_adam_side_by_side.txt
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    mode = sys.argv[3] if len(sys.argv) > 3 else "graph"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    A, obs, _, _ = bench.make_workload("c3", dev)
    B = copy.deepcopy(A)
    pa, pb = list(A.parameters()), list(B.parameters())
    opt_a = bench.make_optimizer(A, False)
    opt_b = bench.make_optimizer(B, True)
    gb = [torch.zeros_like(p) for p in pb]
    for p, g in zip(pb, gb):
        p.grad = g
    rng = zhusuan.DeviceRNG(dev, seed=1)
    one = torch.ones((), device=dev)

    def compute():
        rng.begin_step()
        for p in pa:
            p.grad = None
        loss = A(obs)
        loss.backward(one)
        torch._foreach_copy_(gb, [p.grad for p in pa])
        return loss.detach()

    def update():
        opt_a.step()
        opt_b.step()
    if mode == "graph":
        with zhusuan.device_rng(rng):
            step = zhusuan.GraphedStep(compute, update, rng=rng, warmup=3, optimizer=opt_a)
    else:
        def step():
            with zhusuan.device_rng(rng):
                loss = compute()
                update()
            return loss
    print("c3, lr 1e-3, %s launches; A: FlatAdam, B: torch.optim.Adam(fused, capturable) fed A's gradients" % mode)
    print("%8s %10s | %12s %12s %12s | %s" % ("step", "objective", "d param", "d exp_avg", "d exp_avg_sq", "parameters > 1e-3 apart"))
    for i in range(steps):
        check = (i + 1) % every == 0 or i + 1 in (1, 10, 100)
        if check:
            before_a, before_b = [p.detach().clone() for p in pa], [p.detach().clone() for p in pb]
        loss = step()
        if check:
            with torch.no_grad():
                # THIS step's update, element by element, in units of lr: where the two optimizers' steps differ most
                da = torch.cat([(p - q).reshape(-1) for p, q in zip(pa, before_a)])
                db = torch.cat([(p - q).reshape(-1) for p, q in zip(pb, before_b)])
                dd = (da - db).abs()
                k = int(dd.argmax())
                g_all = torch.cat([g.reshape(-1) for g in gb])
                m_a, v_a = torch.cat([b.exp_avg for b in opt_a.buckets]), torch.cat([b.exp_avg_sq for b in opt_a.buckets])
                m_b = torch.cat([opt_b.state[p]["exp_avg"].reshape(-1) for p in pb])
                v_b = torch.cat([opt_b.state[p]["exp_avg_sq"].reshape(-1) for p in pb])
                big = v_b > 1e-30
                relv = ((v_a - v_b).abs() / v_b.clamp_min(1e-38))[big]
                print("         this step's updates differ by at most %.3e lr (element %d: A %+.6e B %+.6e | g %+.3e | m %+.6e / %+.6e | v %.6e / %.6e);"
                      " exp_avg_sq element-wise: max rel. diff %.2e, %d of %d beyond 1e-3; v == 0: A %d, B %d"
                      % (float(dd[k]) / 1e-3, k, float(da[k]), float(db[k]), float(g_all[k]), float(m_a[k]), float(m_b[k]), float(v_a[k]), float(v_b[k]),
                         float(relv.max()) if relv.numel() else 0.0, int((relv > 1e-3).sum()), int(big.sum()), int((v_a == 0).sum()), int((v_b == 0).sum())))
                fa, fb = torch.cat([p.reshape(-1) for p in pa]), torch.cat([p.reshape(-1) for p in pb])
                ma = torch.cat([b.exp_avg for b in opt_a.buckets])
                va = torch.cat([b.exp_avg_sq for b in opt_a.buckets])
                mb = torch.cat([opt_b.state[p]["exp_avg"].reshape(-1) for p in pb])
                vb = torch.cat([opt_b.state[p]["exp_avg_sq"].reshape(-1) for p in pb])
                rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
                far = int(((fa - fb).abs() > 1e-3).sum())
                print("%8d %10.4f | %12.3e %12.3e %12.3e | %d of %d" % (i + 1, float(loss), rel(fa, fb), rel(ma, mb), rel(va, vb), far, fa.numel()))
                sys.stdout.flush()


if __name__ == "__main__":
    main()
