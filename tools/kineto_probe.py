"""Probe: does torch.profiler (kineto over roctracer) report the kernels of a hipGraph replay, with durations?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd")):
    sys.path.insert(0, p)
import torch
import zhusuan
from examples import iwae
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = iwae.build(50, "vimco", device=dev)
opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True, capturable=True)
rng = zhusuan.DeviceRNG(dev, seed=1)
x = (torch.rand(256, 784, device=dev) < 0.5).float()

def compute():
    rng.begin_step()
    for p in model.parameters():
        p.grad = None
    loss = model({"x": x}); loss.backward()
    return loss.detach()
step = zhusuan.GraphedStep(compute, opt.step, rng=rng, warmup=5)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(20): step()
    torch.cuda.synchronize()
print("profile wall %.1f ms" % (1e3 * (time.perf_counter() - t0)))
agg = {}
for e in prof.events():
    if "k_" in e.name and ("bern" in e.name or "normal" in e.name or "iw_reduce" in e.name):
        d = getattr(e, "device_time", None)
        if d is None: d = getattr(e, "cuda_time", 0.0)
        agg.setdefault(e.name[:70], []).append(d)
for k, v in agg.items():
    print("%-72s n=%3d avg %.2f us min %.2f" % (k, len(v), sum(v) / len(v), min(v)))
print("n events", len(list(prof.events())))
