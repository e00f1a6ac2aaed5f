import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd")): sys.path.insert(0, p)
import torch, zhusuan
from zhusuan import _hip
from examples import iwae
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = iwae.build(50, "sgvb", device=dev); obs = {"x": (torch.rand(256, 784, device=dev) < 0.5).float()}
opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3); rng = zhusuan.DeviceRNG(dev, seed=1)
def step():
    rng.begin_step()
    for p in model.parameters(): p.grad = None
    loss = model(obs); loss.backward(); opt.step()
lib = _hip.lib()
with zhusuan.device_rng(rng):
    for _ in range(5): step()
    torch.cuda.synchronize(); lib.prof_enable(True)
    for _ in range(50): step()
    torch.cuda.synchronize(); lib.prof_enable(False)
for name in ("zs_normal_sample_logprob_f32", "zs_normal_sample_logprob_bwd_f32", "zs_normal_logprob_f32", "zs_normal_logprob_bwd_f32", "zs_normal_logprob_bwd_ksum_f32"):
    q = lib.prof_query(name)
    if q["count"]: print(name, "%.2f us x %.1f/step" % (1e3 * q["total_ms"] / q["count"], q["count"] / 50))
