#!/usr/bin/env python
"""What handing the step a minibatch from HOST memory costs (the reference's loop builds `torch.tensor(x_batch).to(device)` every step,
iwae.py:151-156): a [256, 784] fp32 batch (0.8 MB) copied host -> device, from pageable and from pinned memory, alone and in front
of a graph-replayed training step (copy into the captured input buffer, then replay).  bench.py's `value` has the batch resident in HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))
import torch
import zhusuan
from examples import iwae
dev = torch.device("cuda:0")
B, X = 256, 784
host = [(torch.rand(B, X) < 0.5).float() for _ in range(8)]
pinned = [h.pin_memory() for h in host]
dst = torch.empty(B, X, device=dev)
for label, src, nb in (("pageable", host, False), ("pinned", pinned, True)):
    for _ in range(20):
        dst.copy_(src[0], non_blocking=nb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(400):
        dst.copy_(src[i % 8], non_blocking=nb)
    torch.cuda.synchronize()
    print("host -> device copy of one [256, 784] fp32 minibatch (0.80 MB), %-8s: %6.1f us per copy" % (label, 1e6 * (time.perf_counter() - t0) / 400))
torch.manual_seed(0)
model = iwae.build(50, "vimco", device=dev, dense="fused")
obs = {"x": (torch.rand(B, X, device=dev) < 0.5).float()}
opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
rng = zhusuan.DeviceRNG(dev, seed=1)
def compute():
    rng.begin_step()
    for p in model.parameters():
        p.grad = None
    loss = model(obs); loss.backward(); return loss.detach()
step = zhusuan.GraphedStep(compute, opt.step, rng=rng, inputs=obs)
for label, feed in (("batch resident in HBM", lambda i: step()), ("batch from pinned host memory", lambda i: step(x=pinned[i % 8])),
                    ("batch from pageable host memory", lambda i: step(x=host[i % 8]))):
    for i in range(30):
        feed(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(400):
        feed(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400
    print("graph-replayed IWAE step (B = 256, K = 50, default GEMM picks), %-32s: %.4f ms/step = %.2f M ELBO-evals/s" % (label, 1e3 * dt, 12800 / dt / 1e6))
