"""The particle-batched dense layer of the Bayesian-neural-network caller as one kernel (PL1, include/zs_hip.h).

The reference's example writes the layer out of whole-tensor ops (examples/bayesian_neural_nets/bnn_vi.py:36-48): repeat
the K weight particles over the batch (a 115 MB copy at K = 10, B = 4096 with 50 hidden units), append a column of ones to
the activations, ``matmul``, divide by ``sqrt(n_in + 1)``, ReLU.  ``particle_linear`` is that layer in one launch forward
and one backward; nothing is repeated or concatenated.
"""
import torch

from . import _ops
from .framework.stochastic_tensor import _materialize

__all__ = ['particle_linear', 'particle_mlp', 'particle_rmse', 'Linear', 'Sequential']


def _fits_lds(n_in, n_out, itemsize):
    """The kernel's own admission rule (pl_fits, csrc/zs_layers.hip): a particle's weights plus one 64-row tile of
    activations / gradients must fit 60 KB of LDS."""
    if not (1 <= n_in <= 255 and 1 <= n_out <= 256):
        return False
    lim = 15360 * 4 // itemsize
    odd = lambda v: v | 1
    fwd = 64 * n_in + 4 + n_out * odd(n_in + 1)
    bwd = 64 * n_out + 64 * n_in + 8 + n_out * (n_in + 1)
    return max(fwd, bwd) <= lim


def particle_linear(h, w, relu=False):
    """``out[k, b, :] = act(([h[k, b, :], 1] @ w[k].T) / sqrt(n_in + 1))``.

    :param h: activations ``[K, B, n_in]``, or ``[B, n_in]`` when all particles see the same input (the first layer).
    :param w: one weight matrix per particle, ``[K, n_out, n_in + 1]``; the last column multiplies the appended 1 (bias).
    :param relu: apply ``max(., 0)``.
    :return: ``[K, B, n_out]``.

    Layers that do not fit a workgroup's LDS (n_in > 255, n_out > 256, or more than about 15 000 fp32 weights +
    tile elements) take the equivalent batched-GEMM formulation through torch.
    """
    h, w = _materialize((h, w))          # (node values deferred by zhusuan.skip_discarded_draws)
    K, n_out, n_in1 = w.shape
    n_in = n_in1 - 1
    if _fits_lds(n_in, n_out, w.element_size()):
        return _ops.ParticleLinear.apply(h, w, bool(relu))
    if h.dim() == 2:
        h = h.unsqueeze(0).expand(K, *h.shape)
    out = (torch.bmm(h, w[:, :, :n_in].transpose(1, 2)) + w[:, :, n_in].unsqueeze(1)) / (float(n_in + 1) ** 0.5)
    return torch.relu(out) if relu else out


def _fits_lds_mlp(sizes, itemsize):
    """PM1's admission rule (pm_build, csrc/zs_layers.hip): all layers' weights, the tiles of every layer's input and two
    gradient / activation tiles of the widest layer within 60 KB of LDS, at most 4 layers."""
    L = len(sizes) - 1
    if not (1 <= L <= 4) or any(not (1 <= sizes[l] <= 255 and 1 <= sizes[l + 1] <= 256) for l in range(L)):
        return False
    lim = 15360 * 4 // itemsize
    pad4 = lambda v: (v + 3) & ~3
    odd = lambda v: v | 1
    maxw = max(sizes)
    fwd = 2 * pad4(64 * maxw) + sum(pad4(sizes[l + 1] * odd(sizes[l] + 1)) for l in range(L))
    bwd = 2 * pad4(64 * maxw) + sum(pad4(64 * sizes[l]) for l in range(L)) + sum(pad4(sizes[l + 1] * (sizes[l] + 1)) for l in range(L))
    return max(fwd, bwd) <= lim


def particle_mlp(x, weights):
    """The BNN caller's network (examples/bayesian_neural_nets/bnn_vi.py:27-48): ``h = x``; for every layer
    ``h = ([h, 1] @ w[k].T) / sqrt(n_in + 1)``, ReLU after every layer but the last.

    :param x: inputs ``[B, n_0]`` (shared by the particles) or ``[K, B, n_0]``.
    :param weights: one ``[K, n_out, n_in + 1]`` tensor per layer.
    :return: ``[K, B, n_L]``.

    Networks of up to four layers that fit a workgroup's LDS run as ONE kernel forward and one backward (PM1); anything else
    as a chain of ``particle_linear`` calls (the same values bit for bit)."""
    weights = list(_materialize(tuple(weights)))
    x = _materialize(x)
    if not weights:
        raise ValueError("particle_mlp: at least one layer")
    sizes = [weights[0].shape[2] - 1] + [w.shape[1] for w in weights]
    if len(weights) > 1 and _fits_lds_mlp(sizes, weights[0].element_size()):
        return _ops.ParticleMLP.apply(x, *weights)
    h = x
    for l, w in enumerate(weights):
        h = particle_linear(h, w, relu=l < len(weights) - 1)
    return h


def particle_rmse(pred, y):
    """``sqrt(mean((y - pred.mean(0)) ** 2))``: the error of the particle-mean prediction ``pred`` [K, B] against ``y`` [B], the
    diagnostic the BNN caller evaluates in every forward pass (examples/bayesian_neural_nets/bnn_vi.py:84-87: five torch
    launches), as one kernel (PR1).  Returned without autograd history."""
    return _ops.particle_rmse(*_materialize((pred, y)))


_ACTS = {None: _ops.ACT_NONE, 'relu': _ops.ACT_RELU, 'sigmoid': _ops.ACT_SIGMOID}


class Linear(torch.nn.Linear):
    """``torch.nn.Linear`` (same parameters, same names, same forward GEMM) with the non-GEMM passes of a dense layer folded
    into one kernel each way.  The callers' MLPs of the reference's examples
    (examples/variational_autoencoder/vae_mnist.py:22-28,44-48, iwae.py:40-47,68-75) are stacks of Linear -> ReLU (-> Sigmoid);
    this is glue on the caller's side of the boundary (like ``particle_linear``), not part of the distribution / objective path.

    * backward: the bias gradient is one deterministic column-sum launch (CS1) instead of torch's generic reduction
      (12.4 us per layer for the [12 800, 500] gradients of the IWAE step, seven layers per step);
    * ``activation='relu' | 'sigmoid'`` (or a following ``nn.ReLU`` / ``nn.Sigmoid`` inside a ``zhusuan.Sequential``): the
      ReLU rides in the forward GEMM's epilogue, and the activation's backward is fused with the bias gradient (AB1).
    """

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None, activation=None):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        if activation not in _ACTS:
            raise ValueError("activation: None, 'relu' or 'sigmoid'")
        self.activation = activation

    def forward(self, x, activation='own'):
        act = self.activation if activation == 'own' else activation
        x = _materialize(x)                  # (a node value deferred by zhusuan.skip_discarded_draws)
        kernels = self.weight.dtype in (torch.float32, torch.float64) and x.dtype == self.weight.dtype and \
            not torch.is_autocast_enabled()          # (other precisions: torch's own ops, same results as torch.nn's modules)
        if kernels and (x.requires_grad or self.weight.requires_grad or (self.bias is not None and self.bias.requires_grad)):
            return _ops.DenseLayer.apply(x, self.weight, self.bias, _ACTS[act])
        y = torch.nn.functional.linear(x, self.weight, self.bias)
        return y if act is None else (torch.relu(y) if act == 'relu' else torch.sigmoid(y))


class Sequential(torch.nn.Sequential):
    """``torch.nn.Sequential`` (same children, same indices, same parameter names, slicing included) that runs every
    ``zhusuan.Linear`` followed by an ``nn.ReLU`` / ``nn.Sigmoid`` as one fused layer.  (The activation module of such a pair
    is not called: forward hooks registered on it do not fire.  Use ``torch.nn.Sequential`` where that matters.)"""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, Linear) and m.activation is None and type(nxt) in (torch.nn.ReLU, torch.nn.Sigmoid):
                x = m(x, activation='relu' if type(nxt) is torch.nn.ReLU else 'sigmoid')
                i += 2
            else:
                x = m(x)
                i += 1
        return x
