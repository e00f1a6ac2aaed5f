"""Importance-weighted autoencoder (IWAE / VIMCO) on (synthetic) MNIST-shaped data.

Counterpart of the reference caller examples/variational_autoencoder/iwae.py:34-163: same module
structure / parameter order, K particles on the leading axis, ``reduce_sum_dims=[2]``,
``ImportanceWeightedObjective(generator, variational, axis=0, estimator=...)``.
``fused_logits=True`` drops the decoder's final Sigmoid and hands logits to Bernoulli(logits=...), so the
sigmoid runs inside the log-prob kernel (same p, same +1e-8 formula).
"""
import argparse
import time

import torch
import torch.nn as nn

from zhusuan.framework.bn import BayesianNet
from zhusuan import distributions
from zhusuan.variational.importance_weighted_objective import ImportanceWeightedObjective


def dense_modules(dense):
    """(Linear, Sequential) classes of the callers' MLPs: 'torch' = torch.nn's, as in the reference's example; 'zhusuan' =
    zhusuan.Linear (bias gradient: one column-sum launch) inside torch.nn.Sequential; 'fused' = zhusuan.Linear inside
    zhusuan.Sequential (ReLU in the GEMM's epilogue, activation backward fused with the bias gradient).  Same parameters,
    names and order in all three."""
    if dense == 'torch':
        return nn.Linear, nn.Sequential
    import zhusuan
    if dense == 'zhusuan':
        return zhusuan.Linear, nn.Sequential
    if dense == 'fused':
        return zhusuan.Linear, zhusuan.Sequential
    raise ValueError("dense: 'torch', 'zhusuan' or 'fused'")


def dense_layer(dense):
    return dense_modules(dense)[0]


class Generator(BayesianNet):
    def __init__(self, x_dim, z_dim, n_samples, hidden=500, fused_logits=False, Linear=nn.Linear, Sequential=nn.Sequential):
        super().__init__()
        self.x_dim = x_dim
        self.z_dim = z_dim
        self.n_samples = n_samples
        self.fused_logits = fused_logits
        self.gen_sq = Sequential(
            Linear(z_dim, hidden), nn.ReLU(),
            Linear(hidden, hidden), nn.ReLU(),
            Linear(hidden, x_dim), nn.Sigmoid())
        self._prior = None

    def _prior_params(self, batch_len):
        dev = self.device
        if self._prior is None or self._prior[0].device != dev or self._prior[0].shape[0] != batch_len:
            self._prior = (torch.zeros([batch_len, self.z_dim], device=dev),
                           torch.ones([batch_len, self.z_dim], device=dev))
        return self._prior

    def forward(self, observed):
        self.observe(observed)
        try:
            batch_len = self.observed['z'].shape[1]
        except (KeyError, IndexError):
            batch_len = 64
        mean, std = self._prior_params(batch_len)
        z = self.normal(name="z", mean=mean, std=std, is_reparameterized=False, n_samples=self.n_samples,
                        reduce_mean_dims=None, reduce_sum_dims=[2])
        if self.fused_logits:
            x_logits = self.gen_sq[:-1](z)
            self.cache["x_logits"] = x_logits
            bernoulli = distributions.Bernoulli(logits=x_logits, device=self.device)
        else:
            x_probs = self.gen_sq(z)
            self.cache["x_mean"] = x_probs
            bernoulli = distributions.Bernoulli(probs=x_probs, device=self.device)
        self.sn(bernoulli, name='x', reduce_mean_dims=None, reduce_sum_dims=[2])
        return self


class Variational(BayesianNet):
    def __init__(self, x_dim, z_dim, n_samples, hidden=500, reparameterized=False, Linear=nn.Linear, Sequential=nn.Sequential):
        super().__init__()
        self.x_dim = x_dim
        self.z_dim = z_dim
        self.n_samples = n_samples
        self.reparameterized = reparameterized
        self.output_logits = Sequential(Linear(x_dim, hidden), nn.ReLU(), Linear(hidden, hidden), nn.ReLU())
        self.output_mean = Linear(hidden, z_dim)
        self.output_logstd = Linear(hidden, z_dim)

    def forward(self, observed):
        self.observe(observed)
        x = self.observed['x']
        h = self.output_logits(x)
        z_mean = self.output_mean(h)
        z_std = torch.exp(self.output_logstd(h))
        normal = distributions.Normal(z_mean, z_std, device=self.device, is_reparameterized=self.reparameterized)
        self.sn(normal, name="z", n_samples=self.n_samples, reduce_mean_dims=None, reduce_sum_dims=[2])
        return self


def build(n_samples=50, estimator='vimco', x_dim=784, z_dim=40, hidden=500, device='cuda', fused_logits=False, dense='torch'):
    """`dense`: see dense_modules."""
    Linear, Sequential = dense_modules(dense)
    generator = Generator(x_dim, z_dim, n_samples, hidden, fused_logits, Linear, Sequential)
    variational = Variational(x_dim, z_dim, n_samples, hidden, reparameterized=(estimator == 'sgvb'), Linear=Linear,
                              Sequential=Sequential)
    return ImportanceWeightedObjective(generator, variational, axis=0, estimator=estimator).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--particles', type=int, default=40)
    ap.add_argument('--estimator', default='vimco', choices=['vimco', 'sgvb'])
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--flat-adam', action='store_true',
                    help="zhusuan.optim.FlatAdam: torch.optim.Adam's update as one kernel launch")
    ap.add_argument('--fused-logits', action='store_true')
    ap.add_argument('--dense', default='torch', choices=['torch', 'zhusuan', 'fused'],
                    help="the MLPs' modules: torch.nn's (as the reference), zhusuan.Linear, or zhusuan.Linear in zhusuan.Sequential")
    args = ap.parse_args()
    device = torch.device('cuda')
    model = build(args.particles, args.estimator, device=device, fused_logits=args.fused_logits, dense=args.dense)
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
            print("step %d  surrogate %.4f  IW bound %.4f" % (step + 1, float(loss), float(model.last_iw_bound.mean())))
    torch.cuda.synchronize()
    print("%.1f ELBO-evals/s" % (args.batch * args.particles * args.steps / (time.time() - t0)))


if __name__ == '__main__':
    main()
