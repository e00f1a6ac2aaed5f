"""Importance-weighted objective (IWAE / VIMCO).  Interface of
zhusuan/variational/importance_weighted_objective.py:28-191 of the reference."""
import torch
import torch.nn as nn

from ..framework.stochastic_tensor import StochasticTensor
from .. import _ops
from .elbo import latent_value
from .._shapes import broadcast_shapes

__all__ = ['ImportanceWeightedObjective']

_ERR_VIMCO = ("VIMCO is a multi-sample gradient estimator, size along "
              "`axis` in the objective should be larger than 1.")


class ImportanceWeightedObjective(nn.Module):
    """
    :param generator: BayesianNet p(x, z).
    :param variational: BayesianNet q(z | x).
    :param axis: the particle axis of the log-joint tensors (required).
    :param estimator: 'sgvb' (IWAE, reparameterised latents) or 'vimco' (score function with
        leave-one-out baselines; latents must have is_reparameterized=False).

    After a call, ``last_iw_bound`` holds the detached per-datapoint bound
    ``log_mean_exp(log_w, axis)`` (the returned value is a surrogate for gradients, not the bound).
    """

    def __init__(self, generator, variational, axis=None, estimator='sgvb'):
        super().__init__()
        self.generator = generator
        self.variational = variational
        if axis is None:
            raise ValueError(
                "ImportanceWeightedObjective is a multi-sample objective, "
                "the `axis` argument must be specified.")
        self._axis = axis
        if estimator not in ['sgvb', 'vimco']:
            raise NotImplementedError()
        self.estimator = estimator
        self.last_iw_bound = None

    def log_joint(self, nodes):
        log_joint_ = None
        for n_name in nodes.keys():
            lp = nodes[n_name].log_prob()
            log_joint_ = lp if log_joint_ is None else log_joint_ + lp
        return log_joint_

    def forward(self, observed, reduce_mean=True):
        """importance_weighted_objective.py:79-100."""
        self.variational(observed)
        nodes_q = self.variational.nodes
        _v_inputs = {}
        for k, v in nodes_q.items():
            _v_inputs[k] = latent_value(v)
            if self.estimator == "vimco" and isinstance(v, StochasticTensor) and v.dist.is_reparameterized:
                raise ValueError("with vimco estimator, the is_reparameterized must be false")
        _observed = {**_v_inputs, **observed}
        nodes_p = self.generator(_observed).nodes
        logpxz = self.log_joint(nodes_p)
        logqz = self.log_joint(nodes_q)
        if self.estimator == 'sgvb':
            return self.sgvb(logpxz, logqz, reduce_mean)
        return self.vimco(logpxz, logqz, reduce_mean)

    def _rows(self, logpxz, logqz):
        """Both log-joints as K-fastest [B, K] matrices plus the shape of the non-particle axes.
        Tensors produced by the log-prob kernels already have this layout, so no copy happens."""
        logpxz = torch.as_tensor(logpxz)
        logqz = torch.as_tensor(logqz, device=logpxz.device)
        shape = broadcast_shapes(logpxz.shape, logqz.shape)
        if len(shape) == 0:
            raise ValueError(_ERR_VIMCO)
        axis = self._axis % len(shape)
        K = shape[axis]
        rest = tuple(s for i, s in enumerate(shape) if i != axis)

        def rows(t):
            t = t.expand(shape) if tuple(t.shape) != tuple(shape) else t
            return t.movedim(axis, -1).reshape(-1, K)
        return rows(logpxz), rows(logqz), K, rest

    def sgvb(self, logpxz, logqz, reduce_mean=True):
        """importance_weighted_objective.py:102-132 with compute_iw_term (:16-25) in one kernel."""
        p2, q2, K, rest = self._rows(logpxz, logqz)
        cost_b, bound_b = _ops.IWReduce.apply(p2, q2, _ops.ZS_IW_SGVB)
        self.last_iw_bound = bound_b.reshape(rest)
        if reduce_mean:
            return torch.mean(cost_b)
        return cost_b.reshape(rest)

    def vimco(self, logpxz, logqz, reduce_mean=True):
        """importance_weighted_objective.py:134-191.  Always returns the batch mean, like the reference."""
        try:
            p2, q2, K, rest = self._rows(logpxz, logqz)
        except (IndexError, ZeroDivisionError):
            raise ValueError(_ERR_VIMCO)
        if K < 2:
            raise ValueError(_ERR_VIMCO)
        cost_b, bound_b = _ops.IWReduce.apply(p2, q2, _ops.ZS_IW_VIMCO)
        self.last_iw_bound = bound_b.reshape(rest)
        return cost_b.mean()
