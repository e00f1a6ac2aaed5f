import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch, zhusuan
from examples import iwae, vae_mnist, bnn_vi
dev = torch.device("cuda:0")
for kind in ("iwae", "vae", "bnn"):
    torch.manual_seed(0)
    if kind == "iwae":
        model, obs = iwae.build(50, "vimco", device=dev), {"x": (torch.rand(256, 784, device=dev) < 0.5).float()}
    elif kind == "vae":
        model, obs = vae_mnist.build(512, device=dev), {"x": (torch.rand(512, 784, device=dev) < 0.5).float()}
    else:
        model, obs = bnn_vi.build(n_particles=10, device=dev), {"x": torch.randn(512, 13, device=dev), "y": torch.randn(512, device=dev)}
    opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    rng = zhusuan.DeviceRNG(dev, seed=1)
    def compute():
        rng.begin_step()
        for p in model.parameters(): p.grad = None
        loss = model(obs); loss.backward(); return loss.detach()
    step = zhusuan.GraphedStep(compute, opt.step, rng=rng)
    l0 = float(step()); m0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    for i in range(5000): last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # eager steps too (allocator churn)
    with zhusuan.device_rng(rng):
        for i in range(300): compute(); opt.step()
    torch.cuda.synchronize()
    steps_seen = [b.step.tolist() for b in opt.buckets]
    print(kind, "adam step counters", steps_seen, "tickets", [int(b.ticket.item()) for b in opt.buckets])
    print(kind, "loss %.2f -> %.2f" % (l0, float(last)), "ms/step %.4f" % (1e3 * dt / 5000), "mem delta %d B" % (torch.cuda.memory_allocated() - m0), "finite", bool(torch.isfinite(last)))
