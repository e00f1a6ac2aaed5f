"""The six families the reference wraps around ``torch.distributions`` -- Beta, Exponential, Gamma, Laplace, Poisson,
StudentT (zhusuan/distributions/{beta,exponential,gamma,laplace,poisson,studentT}.py) -- as thin PASS-THROUGHS.

They are off the variational-inference hot path named by BASELINE.json: no HIP kernel
is written for them and none is claimed.  They exist so that model code written against the reference keeps working after
the switch: sampling and log-prob are exactly the reference's ``torch.distributions`` calls (on whatever device the
parameters live on), with the reference's conventions -- parameters repeated along a leading sample axis, never
reparameterised (``sample()``, not ``rsample()``), ``sample_cache``, the group sum of ``Distribution.log_prob``.
One generic implementation instead of six files.  ``FlowDistribution`` needs ``zhusuan.invertible`` and stays out.
"""
import warnings

import torch

from .base import Distribution
from .utils import assert_same_log_float_dtype, check_broadcast
from .. import _hip
from .._shapes import broadcast_shapes

__all__ = ['Beta', 'Exponential', 'Gamma', 'Laplace', 'Poisson', 'StudentT', 'FlowDistribution']

_INT2FLOAT = {torch.int8: torch.float16, torch.int16: torch.float16, torch.int32: torch.float32, torch.int64: torch.float64,
              torch.uint8: torch.float16}


def _repeat_like_reference(p, n_samples, anchor_ndim):
    """``p.repeat([n_samples, 1, ..., 1])`` with one 1 per axis of the family's ANCHOR parameter (e.g. beta.py:50-54:
    ``_len = len(self._alpha.shape)`` serves alpha and beta alike): a leading sample axis, lower-rank parameters padded."""
    return p.repeat([n_samples] + [1] * anchor_ndim)


class _TorchFamily(Distribution):
    """Generic pass-through: subclasses name the torch class, the parameters (in the reference's argument order) and
    their defaults."""
    _torch_cls = None
    _params = ()              # ((name, default or _REQUIRED), ...)
    _int_rate_ok = False      # Poisson: an integer rate is converted with a warning (poisson.py:29-31)
    _anchor = None            # the parameter whose rank decides the repeat pattern and the "has a sample axis" test

    def __init__(self, *args, dtype=None, is_continuous=True, group_ndims=0, device=None, **kwargs):
        names = [n for n, _ in self._params]
        if len(args) > len(names):
            raise TypeError("%s takes at most %d positional parameters" % (type(self).__name__, len(names)))
        given = dict(zip(names, args))
        for n, default in self._params:
            if n in kwargs:
                if n in given:
                    raise TypeError("%s got multiple values for argument '%s'" % (type(self).__name__, n))
                given[n] = kwargs.pop(n)
            elif n not in given:
                if default is _REQUIRED:
                    raise TypeError("%s missing required argument '%s'" % (type(self).__name__, n))
                given[n] = default
        device = _hip.resolve_device(device, *[given[n] for n in names])
        vals = []
        for n in names:
            v = torch.as_tensor(given[n], dtype=dtype).to(device)
            if self._int_rate_ok and v.dtype in _INT2FLOAT:
                warnings.warn("the tensor dtype convert %s to  %s" % (v.dtype, _INT2FLOAT[v.dtype]))
                v = torch.as_tensor(v, dtype=_INT2FLOAT[v.dtype])
            vals.append(v)
        for a, b in zip(vals, vals[1:]):
            check_broadcast(a, b)
        dtype = assert_same_log_float_dtype([(v, "%s.%s" % (type(self).__name__, n)) for n, v in zip(names, vals)])
        for n, v in zip(names, vals):
            setattr(self, "_" + n, v)
        # the reparameterisation trick is not applied for these families (e.g. exponential.py:27-29)
        super(_TorchFamily, self).__init__(dtype, is_continuous, is_reparameterized=False, group_ndims=group_ndims,
                                           device=device, **kwargs)

    def _values(self):
        return [getattr(self, "_" + n) for n, _ in self._params]

    def _batch_shape(self):
        return torch.Size(broadcast_shapes(*[v.shape for v in self._values()]))

    def _sample(self, n_samples=1, **kwargs):
        vals = self._values()
        if n_samples > 1:
            nd = getattr(self, "_" + self._anchor).dim()
            vals = [_repeat_like_reference(v, n_samples, nd) for v in vals]
        s = self._torch_cls(*vals).sample()
        self.sample_cache = s
        return s

    def _log_prob_sum(self, given=None, n_fold=0):
        x = self.sample_cache if given is None else given
        if x is None:
            raise RuntimeError("%s.log_prob(None) needs a cached sample: call sample() first" % type(self).__name__)
        vals = self._values()
        nd = getattr(self, "_" + self._anchor).dim()
        if x.dim() > nd:
            vals = [_repeat_like_reference(v, x.shape[0], nd) for v in vals]
        lp = self._torch_cls(*vals).log_prob(x)
        if n_fold > 0:
            lp = lp.sum(tuple(range(lp.dim() - n_fold, lp.dim())))
        return lp


_REQUIRED = object()


def _family(name, torch_cls, params, doc, int_rate_ok=False, anchor=None):
    ns = {"_torch_cls": torch_cls, "_params": tuple(params), "_int_rate_ok": int_rate_ok, "__doc__": doc,
          "_anchor": anchor or params[0][0]}
    for n, _ in params:
        ns[n] = property(lambda self, _n=n: getattr(self, "_" + _n))
    return type(name, (_TorchFamily,), ns)


Beta = _family('Beta', torch.distributions.beta.Beta, [("alpha", _REQUIRED), ("beta", _REQUIRED)],
               "Beta(alpha, beta): pass-through of torch.distributions.Beta (zhusuan/distributions/beta.py).")
Exponential = _family('Exponential', torch.distributions.exponential.Exponential, [("rate", _REQUIRED)],
                      "Exponential(rate): pass-through of torch.distributions.Exponential (exponential.py).")
Gamma = _family('Gamma', torch.distributions.gamma.Gamma, [("alpha", _REQUIRED), ("beta", _REQUIRED)],
                "Gamma(alpha, beta): concentration alpha, rate beta; pass-through of torch.distributions.Gamma (gamma.py).")
Laplace = _family('Laplace', torch.distributions.laplace.Laplace, [("loc", _REQUIRED), ("scale", _REQUIRED)],
                  "Laplace(loc, scale): pass-through of torch.distributions.Laplace (laplace.py).", anchor="loc")
Poisson = _family('Poisson', torch.distributions.poisson.Poisson, [("rate", _REQUIRED)],
                  "Poisson(rate): pass-through of torch.distributions.Poisson (poisson.py).", int_rate_ok=True)
StudentT = _family('StudentT', torch.distributions.studentT.StudentT, [("df", _REQUIRED), ("loc", 0.), ("scale", 1.)],
                   "StudentT(df, loc=0, scale=1): pass-through of torch.distributions.StudentT (studentT.py).", anchor="loc")


class FlowDistribution(Distribution):
    """Not part of the MI355X build: needs ``zhusuan.invertible`` (normalising flows, outside the hot path)."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "zhusuan.distributions.FlowDistribution is outside the variational-inference hot path of the MI355X build "
            "(it depends on zhusuan.invertible)")
