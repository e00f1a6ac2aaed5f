"""Bayesian neural network regression with mean-field VI (SGVB ELBO) on synthetic UCI-sized data.

Counterpart of the reference caller examples/bayesian_neural_nets/bnn_vi.py:16-99: a prior node and a
variational node per weight matrix (``group_ndims=2``, K particles, ``reduce_mean_dims=[0]``), a Normal
likelihood with ``multiplier`` = training-set size.  The particle-batched network -- in the reference, per layer,
``w.repeat([1, B, 1, 1])``, a column of ones appended to h, ``matmul``, ``/ sqrt(n_in + 1)``, ReLU (bnn_vi.py:27-48) --
runs ALL its layers as ONE kernel each way with ``layer='fused'`` (default: ``zhusuan.particle_mlp``, PM1 of include/zs_hip.h)
and one kernel per layer and direction with ``layer='per_layer'`` (``zhusuan.particle_linear``, PL1); ``layer='bmm'`` is round 2's
batched-GEMM formulation (cat + bmm + div + relu), ``layer='materialize'`` reproduces the reference's op sequence.
"""
import argparse
import math
import time

import torch

import zhusuan
from zhusuan.framework.bn import BayesianNet
from zhusuan.variational.elbo import ELBO


class Net(BayesianNet):
    def __init__(self, layer_sizes, n_particles, multiplier=456, materialize=False, layer=None):
        super().__init__()
        self.layer_sizes = layer_sizes
        self.n_particles = n_particles
        self.multiplier = multiplier
        self.layer = layer or ('materialize' if materialize else 'fused')
        if self.layer not in ('fused', 'per_layer', 'bmm', 'materialize'):
            raise ValueError("layer: 'fused', 'per_layer', 'bmm' or 'materialize'")
        self.y_logstd = torch.nn.Parameter(torch.zeros([1], dtype=torch.float32))
        self._priors = None
        self._ones = {}

    def _prior_params(self):
        dev = self.device
        if self._priors is None or self._priors[0][0].device != dev:
            self._priors = [(torch.zeros([n_out, n_in + 1], device=dev), torch.ones([n_out, n_in + 1], device=dev))
                            for n_in, n_out in zip(self.layer_sizes[:-1], self.layer_sizes[1:])]
        return self._priors

    def forward(self, observed):
        self.observe(observed)
        x = self.observed['x']
        K = self.n_particles
        h = x if self.layer in ('fused', 'per_layer') else x.unsqueeze(0).expand(K, *x.shape)      # (the reference repeats x K times, :27)
        ws = []
        batch_size = x.shape[0]
        priors = self._prior_params()
        n_layers = len(self.layer_sizes) - 1
        for i in range(n_layers):
            w = self.normal(name='w' + str(i), mean=priors[i][0], std=priors[i][1], group_ndims=2,
                            n_samples=K, reduce_mean_dims=[0])
            last = i == n_layers - 1
            if self.layer == 'fused':                 # the whole network below, in one launch (PM1)
                ws.append(w)
                continue
            if self.layer == 'per_layer':             # one launch per layer (PL1)
                h = zhusuan.particle_linear(h, w, relu=not last)
                continue
            key = (tuple(h.shape[:-1]), h.device, h.dtype)
            ones = self._ones.get(key)       # the bias column is a constant: built once, not once per layer and step
            if ones is None:
                ones = self._ones[key] = torch.ones([*h.shape[:-1], 1], device=h.device, dtype=h.dtype)
            h = torch.cat((h, ones), -1)
            scale = math.sqrt(h.shape[2])      # host scalar: no H2D copy inside a captured step (bnn_vi.py:42)
            if self.layer == 'materialize':
                wr = torch.unsqueeze(w, 1).repeat([1, batch_size, 1, 1])
                h = torch.squeeze(torch.matmul(wr, torch.unsqueeze(h, -1)), -1) / scale
            else:
                h = torch.bmm(h, w.transpose(1, 2)) / scale
            if not last:
                h = torch.relu(h)
        if self.layer == 'fused':
            h = zhusuan.particle_mlp(x, ws)
        y_mean = torch.squeeze(h, 2)
        y = self.observed['y']
        if self.layer in ('fused', 'per_layer'):
            # sqrt(mean((y - mean(y_mean, 0))^2)) (bnn_vi.py:84-87) as one launch
            self.cache['rmse'] = zhusuan.particle_rmse(y_mean, y)
        else:
            y_pred = torch.mean(y_mean, 0)
            # ... as norm / sqrt(B): one reduction launch instead of pow + mean + sqrt
            self.cache['rmse'] = torch.linalg.vector_norm(y - y_pred) * (1.0 / math.sqrt(max(y.numel(), 1)))
        self.normal(name='y', mean=y_mean, logstd=self.y_logstd, reduce_mean_dims=[0, 1],
                    multiplier=self.multiplier)
        return self


class Variational(BayesianNet):
    def __init__(self, layer_sizes, n_particles):
        super().__init__()
        self.layer_sizes = layer_sizes
        self.n_particles = n_particles
        means, logstds = [], []
        for n_in, n_out in zip(layer_sizes[:-1], layer_sizes[1:]):
            means.append(torch.nn.Parameter(torch.zeros([n_out, n_in + 1], dtype=torch.float32)))
            logstds.append(torch.nn.Parameter(torch.zeros([n_out, n_in + 1], dtype=torch.float32)))
        self.w_means = torch.nn.ParameterList(means)
        self.w_logstds = torch.nn.ParameterList(logstds)

    def forward(self, observed):
        self.observe(observed)
        for i in range(len(self.layer_sizes) - 1):
            self.normal(name='w' + str(i), mean=self.w_means[i], logstd=self.w_logstds[i], group_ndims=2,
                        n_samples=self.n_particles, reduce_mean_dims=[0])
        return self


def build(layer_sizes=(13, 50, 1), n_particles=10, multiplier=456, device='cuda', materialize=False, layer=None):
    net = Net(list(layer_sizes), n_particles, multiplier, materialize, layer)
    variational = Variational(list(layer_sizes), n_particles)
    return ELBO(net, variational).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=114)
    ap.add_argument('--particles', type=int, default=512)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--flat-adam', action='store_true',
                    help="zhusuan.optim.FlatAdam: torch.optim.Adam's update as one kernel launch")
    ap.add_argument('--layer', default='fused', choices=['fused', 'per_layer', 'bmm', 'materialize'],
                    help="the particle-batched network: one kernel each way (PM1), one per layer (PL1), round 2's batched GEMMs, "
                         "or the reference's op sequence")
    args = ap.parse_args()
    device = torch.device('cuda')
    model = build(n_particles=args.particles, device=device, layer=args.layer)
    if args.flat_adam:
        import zhusuan
        opt = zhusuan.optim.FlatAdam(model.parameters(), lr=1e-3)
    else:
        opt = torch.optim.Adam(model.parameters(), 1e-3)          # as the reference's example
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.batch, 13, generator=g).to(device)
    y = torch.randn(args.batch, generator=g).to(device)
    t0 = time.time()
    for step in range(args.steps):
        loss = model({'x': x, 'y': y})
        opt.zero_grad()
        loss.backward()
        opt.step()
        if (step + 1) % 50 == 0:
            print("step %d  -ELBO %.4f  rmse %.4f" % (step + 1, float(loss), float(model.generator.cache['rmse'])))
    torch.cuda.synchronize()
    print("%.1f ELBO-evals/s" % (args.batch * args.particles * args.steps / (time.time() - t0)))


if __name__ == '__main__':
    main()
