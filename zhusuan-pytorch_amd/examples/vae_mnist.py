"""VAE on (synthetic) MNIST-shaped data with the SGVB ELBO.

Counterpart of the reference caller examples/variational_autoencoder/vae_mnist.py:16-126: same module
structure and parameter order (so weights are interchangeable), same node definitions, same objective.
Differences: the N(0, 1) prior parameters are created on the model's device instead of on the CPU
(the reference copies them host->device every step, vae_mnist.py:33-34), the data are synthetic
Bernoulli(0.5) bits (the reference downloads MNIST), and ``hidden`` is a constructor argument.
"""
import argparse
import time

import torch
import torch.nn as nn

from zhusuan.framework.bn import BayesianNet
from zhusuan.variational.elbo import ELBO


class Generator(BayesianNet):
    def __init__(self, x_dim, z_dim, batch_size, hidden=500, Linear=nn.Linear, Sequential=nn.Sequential):
        super().__init__()
        self.x_dim = x_dim
        self.z_dim = z_dim
        self.batch_size = batch_size
        self.sequential = Sequential(
            Linear(z_dim, hidden), nn.ReLU(),
            Linear(hidden, hidden), nn.ReLU(),
            Linear(hidden, x_dim), nn.Sigmoid())
        self._prior = None

    def _prior_params(self):
        dev = self.device
        if self._prior is None or self._prior[0].device != dev or self._prior[0].shape[0] != self.batch_size:
            self._prior = (torch.zeros([self.batch_size, self.z_dim], device=dev),
                           torch.ones([self.batch_size, self.z_dim], device=dev))
        return self._prior

    def forward(self, observed):
        self.observe(observed)
        mean, std = self._prior_params()
        z = self.normal(name='z', mean=mean, std=std, reduce_mean_dims=[0], reduce_sum_dims=[1])
        x_probs = self.sequential(z)
        self.cache['x_mean'] = x_probs
        self.bernoulli(name='x', probs=x_probs, reduce_mean_dims=[0], reduce_sum_dims=[1])
        return self


class Variational(BayesianNet):
    def __init__(self, x_dim, z_dim, batch_size, hidden=500, Linear=nn.Linear, Sequential=nn.Sequential):
        super().__init__()
        self.x_dim = x_dim
        self.z_dim = z_dim
        self.batch_size = batch_size
        self.sq = Sequential(Linear(x_dim, hidden), nn.ReLU(), Linear(hidden, hidden), nn.ReLU())
        self.fc3 = Linear(hidden, z_dim)
        self.fc4 = Linear(hidden, z_dim)

    def forward(self, observed):
        self.observe(observed)
        x = self.observed['x']
        h = self.sq(x)
        z_mean = self.fc3(h)
        z_std = torch.exp(self.fc4(h))
        self.normal(name='z', mean=z_mean, std=z_std, reduce_mean_dims=[0], reduce_sum_dims=[1])
        return self


def build(batch_size=64, x_dim=784, z_dim=40, hidden=500, device='cuda', dense='torch'):
    """`dense`: 'torch' | 'zhusuan' | 'fused' (examples/iwae.py: dense_modules)."""
    from .iwae import dense_modules
    Linear, Sequential = dense_modules(dense)
    generator = Generator(x_dim, z_dim, batch_size, hidden, Linear, Sequential)
    variational = Variational(x_dim, z_dim, batch_size, hidden, Linear, Sequential)
    return ELBO(generator, variational).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--flat-adam', action='store_true',
                    help="zhusuan.optim.FlatAdam: torch.optim.Adam's update as one kernel launch")
    ap.add_argument('--dense', default='torch', choices=['torch', 'zhusuan', 'fused'],
                    help="the MLPs' modules: torch.nn's (as the reference), zhusuan.Linear, or zhusuan.Linear in zhusuan.Sequential")
    args = ap.parse_args()
    device = torch.device('cuda')
    model = build(args.batch, device=device, dense=args.dense)
    if args.flat_adam:
        import zhusuan
        opt = zhusuan.optim.FlatAdam(model.parameters(), lr=args.lr)
    else:
        opt = torch.optim.Adam(model.parameters(), args.lr)          # as the reference's example
    g = torch.Generator().manual_seed(1234)
    x_all = (torch.rand(args.batch * 32, 784, generator=g) < 0.5).float().to(device)
    t0 = time.time()
    for step in range(args.steps):
        i = (step % 32) * args.batch
        loss = model({'x': x_all[i:i + args.batch]})
        opt.zero_grad()
        loss.backward()
        opt.step()
        if (step + 1) % 50 == 0:
            print("step %d  loss %.4f" % (step + 1, float(loss)))
    torch.cuda.synchronize()
    print("%.1f ELBO-evals/s" % (args.batch * args.steps / (time.time() - t0)))


if __name__ == '__main__':
    main()
