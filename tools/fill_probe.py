"""Which aten ops launch the fill / memset kernels of one IWAE-VIMCO training step?  (developer probe)"""
import os, sys
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

def step():
    rng.begin_step()
    for p in model.parameters():
        p.grad = None
    loss = model({"x": x}); loss.backward(); opt.step()
with zhusuan.device_rng(rng):
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::ones_like", "aten::full", "aten::ones", "aten::new_zeros"):
        par = e.cpu_parent.name if e.cpu_parent is not None else None
        st = [s for s in (e.stack or []) if "/repo/" in s or "optim" in s][:3]
        print("%-18s parent=%-40s shapes=%s %s" % (e.name, par, str(e.input_shapes)[:50], st))
