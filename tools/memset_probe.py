#!/usr/bin/env python
"""Where do the `Memset (Device)` nodes of a training step come from?  Counts them (torch.profiler) for the VAE config's
full step, for the bare nn.Linear stacks of its two MLPs (forward + backward, same shapes) and for the objective's
distribution kernels alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zhusuan-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torch.profiler import profile, ProfilerActivity

import zhusuan
from examples import vae_mnist

dev = torch.device("cuda:0")


def count(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    acc = {}
    for e in prof.events():
        if "emset" in e.name or "Fill" in e.name:
            acc[e.name[:60]] = acc.get(e.name[:60], 0) + 1
    return {k: v / n for k, v in acc.items()}


B = 512
model = vae_mnist.build(B, device=dev)
x = (torch.rand(B, 784, device=dev) < 0.5).float()
opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)


def full():
    for p in model.parameters():
        p.grad = None
    loss = model({"x": x})
    loss.backward()
    opt.step()


print("full step              ", count(full))
enc = torch.nn.Sequential(torch.nn.Linear(784, 500), torch.nn.ReLU(), torch.nn.Linear(500, 500), torch.nn.ReLU(), torch.nn.Linear(500, 80)).to(dev)
dec = torch.nn.Sequential(torch.nn.Linear(40, 500), torch.nn.ReLU(), torch.nn.Linear(500, 500), torch.nn.ReLU(), torch.nn.Linear(500, 784),
                          torch.nn.Sigmoid()).to(dev)


def mlps():
    for m in (enc, dec):
        for p in m.parameters():
            p.grad = None
    h = enc(x)
    out = dec(h[:, :40])
    (out.sum() + h.sum()).backward()


print("bare nn.Linear stacks  ", count(mlps))
mu = torch.randn(B, 40, device=dev, requires_grad=True)
sd = (torch.rand(B, 40, device=dev) + 0.5).requires_grad_()
pr = (torch.rand(B, 784, device=dev) * 0.9 + 0.05).requires_grad_()


def kernels_only():
    q = zhusuan.distributions.Normal(mean=mu, std=sd, group_ndims=1)
    z = q.sample()
    lq = q.log_prob(z)
    lp = zhusuan.distributions.Normal(mean=torch.zeros_like(mu), std=torch.ones_like(sd), group_ndims=1).log_prob(z)
    lx = zhusuan.distributions.Bernoulli(probs=pr, group_ndims=1).log_prob(x)
    (lq.mean() + lp.mean() + lx.mean()).backward()
    mu.grad = sd.grad = pr.grad = None


print("distribution kernels   ", count(kernels_only))
