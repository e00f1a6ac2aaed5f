"""Univariate Bernoulli.  API of zhusuan/distributions/bernoulli.py:10-98 of the reference; log_prob is the
fused row-sum kernel K3, sampling is K5 (include/zs_hip.h)."""
import torch

from .base import Distribution
from .utils import assert_same_log_float_dtype
from .. import _hip, _ops, _rng
from .._shapes import value_shape

__all__ = ['Bernoulli']


class Bernoulli(Distribution):
    _nonreparam_draw_has_zero_grad = True      # see Distribution
    """
    :param logits / probs: exactly one of them (ValueError otherwise, bernoulli.py:39-42).
        From ``probs`` the log-mass is ``x*log(p+1e-8) + (1-x)*log(1-p+1e-8)`` (bernoulli.py:94);
        from ``logits`` the same formula is applied to ``p = sigmoid(logits)`` (bernoulli.py:50), with
        the sigmoid evaluated inside the kernel so p never round-trips through HBM.
        The derived parameter (``logits`` from probs, ``probs`` from logits) is computed on first
        access instead of eagerly (the reference spends three extra passes over p in the constructor,
        bernoulli.py:45).
    """

    def __init__(self,
                 logits=None,
                 probs=None,
                 dtype=None,
                 is_continuous=False,
                 group_ndims=0,
                 device=None,
                 **kwargs):
        device = _hip.resolve_device(device, logits, probs)
        if (logits is None) == (probs is None):
            raise ValueError(
                "Either `probs` or `logits` should be passed. It is not allowed "
                "that both are specified or both are not.")
        elif logits is None:
            self._probs = torch.as_tensor(probs, dtype=dtype).to(device)
            self._logits = None
            self._from_logits = False
            dtype = assert_same_log_float_dtype([(self._probs, "Bernoulli.probs")])
        else:
            self._logits = torch.as_tensor(logits, dtype=dtype).to(device)
            self._probs = None
            self._from_logits = True
            dtype = assert_same_log_float_dtype([(self._logits, "Bernoulli.logits")])
        super(Bernoulli, self).__init__(dtype,
                                        is_continuous,
                                        is_reparameterized=False,
                                        group_ndims=group_ndims,
                                        device=device,
                                        **kwargs)

    @property
    def probs(self):
        if self._probs is None:
            self._probs = torch.sigmoid(self._logits)
        return self._probs

    @property
    def logits(self):
        if self._logits is None:
            p = self._probs
            self._logits = torch.log(p / (torch.ones_like(p) - p))
        return self._logits

    def _param(self):
        return self._logits if self._from_logits else self._probs

    def _batch_shape(self):
        return self._param().shape

    def _sample(self, n_samples=1, **kwargs):
        """bernoulli.py:72-82: ``torch.bernoulli(probs)`` -> u < p with u from the Philox stream."""
        K = int(n_samples)
        p = self.probs.contiguous()
        shape = ((K,) if K > 1 else ()) + tuple(p.shape)
        if _rng.reference_stream_active():       # zhusuan.reference_rng(): torch.bernoulli on the host, as the reference
            s = torch.bernoulli(p.detach().cpu().expand(shape)).to(p.device)
            self.sample_cache = s
            return s
        seed, call, rng_state = _rng.next_call(p.device)
        s = _ops.bernoulli_sample(p, max(p.numel(), 1), shape, seed, call, rng_state)
        self.sample_cache = s
        return s

    def _lj_term(self, given, n_fold):
        """This node's contribution to a scalar log-joint as a term of the one-launch objective (LJ1), see Normal."""
        x = self.sample_cache if given is None else given
        if x is None:
            raise RuntimeError("Bernoulli.log_prob(None) needs a cached sample: call sample() first")
        par = self._param()
        x = torch.as_tensor(x, dtype=self._dtype).to(par.device)
        full = value_shape(x.shape, par.dim(), par.shape)
        px, Px = _ops.periodic_operand(x, full)
        pp, Pp = _ops.periodic_operand(par, full)
        fam = _hip.LJ_BERNOULLI_LOGITS if self._from_logits else _hip.LJ_BERNOULLI
        return fam, (px, pp, None), (Px, Pp, 1), _ops._prod(full)

    def _log_prob_sum(self, given=None, n_fold=0):
        """bernoulli.py:84-95 (+ trailing sum over `n_fold` axes)."""
        x = self.sample_cache if given is None else given
        if x is None:
            raise RuntimeError("Bernoulli.log_prob(None) needs a cached sample: call sample() first")
        par = self._param()
        x = torch.as_tensor(x, dtype=self._dtype).to(par.device)
        full = value_shape(x.shape, par.dim(), par.shape)
        if n_fold > len(full):
            raise ValueError("cannot sum %d trailing axes of a result of shape %s" % (n_fold, full))
        p_full = par if tuple(par.shape) == full else par.expand(full)
        p_full = p_full.contiguous()
        px, Px = _ops.periodic_operand(x, full)
        return _ops.BernoulliLogProb.apply(p_full, px, n_fold, Px, True, self._from_logits)
