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
import sys as _s
KIND = _s.argv[1] if len(_s.argv) > 1 else "iwae"
from examples import vae_mnist, bnn_vi
model = iwae.build(50, "vimco", device=dev) if KIND == "iwae" else (vae_mnist.build(512, device=dev) if KIND == "vae" else bnn_vi.build(n_particles=10, device=dev))
opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True, capturable=True)
rng = zhusuan.DeviceRNG(dev, seed=1)
x = (torch.rand(256 if KIND == "iwae" else 512, 784, device=dev) < 0.5).float()
obs = {"x": x} if KIND != "bnn" else {"x": torch.randn(512, 13, device=dev), "y": torch.randn(512, device=dev)}

def compute():
    rng.begin_step()
    for p in model.parameters():
        p.grad = None
    loss = model(obs); loss.backward()
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
other = {}
for e in prof.events():
    if getattr(e, "device_type", None) is not None and "DeviceType.CUDA" in str(e.device_type):
        d = getattr(e, "device_time", 0.0) or 0.0
        other.setdefault(e.name[:90], []).append(d)
tot = sum(sum(v) for v in other.values())
for k, v in sorted(other.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print("%5.1f%% %6.1f/step %7.2f us  %s" % (100 * sum(v) / tot, len(v) / 20.0, sum(v) / len(v), k))
print("device time per step %.1f us" % (tot / 20.0))
print("n events", len(list(prof.events())))
