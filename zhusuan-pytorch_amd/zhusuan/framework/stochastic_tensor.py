"""StochasticTensor: a named node = (BayesianNet, Distribution, n_samples, reduction kwargs).
Interface of zhusuan/framework/stochastic_tensor.py:5-181 of the reference."""
import torch

from .._shapes import broadcast_shapes

__all__ = ['StochasticTensor']


def _norm_dims(dims, nd):
    """Axes of a reduction made non-negative.  An axis outside [-nd, nd) raises IndexError like the reference's
    ``torch.mean`` / ``torch.sum`` (stochastic_tensor.py:162-165) instead of silently wrapping onto another axis."""
    if not dims:
        return []
    out = []
    for d in dims:
        d = int(d)
        lo, hi = (-nd, nd - 1) if nd else (-1, 0)
        if not lo <= d <= hi:
            raise IndexError("Dimension out of range (expected to be in range of [%d, %d], but got %d)" % (lo, hi, d))
        out.append(d % nd if nd else 0)
    return out


class StochasticTensor(object):
    """
    :param bn: the owning BayesianNet.
    :param name: unique node name.
    :param dist: a Distribution instance.
    :param n_samples: number of particles drawn by ``.tensor`` (None = one sample, no leading axis).
    :param reduce_mean_dims / reduce_sum_dims / multiplier: post-processing of ``log_prob()``
        (stochastic_tensor.py:58-60,160-181).
    """

    def __init__(self, bn, name, dist, observation=None, n_samples=None, **kwargs):
        self._bn = bn
        self._name = name
        self._dist = dist
        self._dtype = dist.dtype
        self._n_samples = n_samples
        self._observation = observation
        self._check_observation(observation)
        self._reduce_mean_dims = kwargs.get("reduce_mean_dims", None)
        self._reduce_sum_dims = kwargs.get("reduce_sum_dims", None)
        self._multiplier = kwargs.get("multiplier", None)

    def _check_observation(self, observation):
        """stochastic_tensor.py:62-68: an observation handed to the constructor is cast to the node's dtype."""
        if observation is None:
            return None
        if observation.dtype != self.dtype:
            observation = torch.as_tensor(observation, dtype=self.dtype)
        return observation

    @property
    def bn(self):
        return self._bn

    @property
    def name(self):
        return self._name

    @property
    def dtype(self):
        return self._dtype

    @property
    def dist(self):
        return self._dist

    def is_observed(self):
        """stochastic_tensor.py:106-112: whether an observation was handed to the CONSTRUCTOR (``.tensor`` looks the
        name up in ``bn.observed`` instead, :121 -- the two differ and both are kept as in the reference)."""
        return self._observation is not None

    @property
    def tensor(self):
        """Observed value if the node is observed (also primes ``dist.sample_cache``,
        stochastic_tensor.py:122-124), otherwise a FRESH sample on every access (:126)."""
        if self._name in self._bn.observed.keys():
            self._dist.sample_cache = self._bn.observed[self._name]
            return self._bn.observed[self._name]
        return self._dist.sample(n_samples=self._n_samples)

    def sample(self, force=False):
        if self._name in self._bn.observed.keys() and not force:
            self._dist.sample_cache = self._bn.observed[self._name]
            return self._bn.observed[self._name]
        return self._dist.sample(n_samples=self._n_samples)

    @property
    def shape(self):
        return self.tensor.shape

    def get_shape(self):
        return self.shape

    def _reduction_plan(self, sample=None):
        """(value, ndim of dist.log_prob, mean dims, sum dims, trailing sum dims folded into the kernel)."""
        dist = self._dist
        g = dist.group_ndims
        x = dist.sample_cache if sample is None else sample
        if x is None:
            raise RuntimeError("node '%s' has no value yet" % self._name)
        full = tuple(broadcast_shapes(tuple(torch.as_tensor(x).shape), tuple(dist.batch_shape)))
        nd = len(full) - g  # ndim of dist.log_prob(x)
        mean_dims = _norm_dims(self._reduce_mean_dims, nd)
        sum_dims = _norm_dims(self._reduce_sum_dims, nd)
        extra = 0
        while nd - 1 - extra >= 0 and (nd - 1 - extra) in sum_dims and (nd - 1 - extra) not in mean_dims:
            extra += 1
        return full, nd, mean_dims, sum_dims, extra

    def _scalar_term(self, sample=None, rows=True):
        """When the node's reductions collapse EVERY axis (the VAE / BNN callers), ``log_prob()`` is
        ``coef * rows.sum()``: returns (rows, coef) so that the objective can fold all nodes into one launch
        (zs_scalar_objective); None otherwise.  Means and sums over distinct axes commute, so
        coef = multiplier / prod(sizes of the mean axes).  ``rows=False`` only answers the question (no kernel)."""
        full, nd, mean_dims, sum_dims, extra = self._reduction_plan(sample)
        if nd == 0 or set(mean_dims) | set(sum_dims) != set(range(nd)) or set(mean_dims) & set(sum_dims):
            return None
        coef = 1.0
        for d in mean_dims:
            coef /= float(full[d])
        if self._multiplier:
            coef *= float(self._multiplier)
        if not rows:
            return None, coef
        return self._dist._log_prob_sum(sample, self._dist.group_ndims + extra), coef

    def log_prob(self, sample=None):
        """stochastic_tensor.py:160-181: dist.log_prob -> mean over reduce_mean_dims -> sum over
        reduce_sum_dims -> drop those axes -> * multiplier.

        Trailing axes that are only summed are folded into the log-prob kernel's row sum (together with
        the distribution's group_ndims axes); the remaining reductions act on the already reduced,
        small tensor.  Sums and means over distinct axes commute, so the value is the reference's up to
        fp32 summation order."""
        dist = self._dist
        g = dist.group_ndims
        full, nd, mean_dims, sum_dims, extra = self._reduction_plan(sample)
        lp = dist._log_prob_sum(sample, g + extra)
        rest_sum = [d for d in sum_dims if d < nd - extra]
        if mean_dims:
            lp = torch.mean(lp, mean_dims, keepdim=True)
        if rest_sum:
            lp = torch.sum(lp, rest_sum, keepdim=True)
        dims = sorted(set(mean_dims) | set(rest_sum), reverse=True)
        for d in dims:
            lp = torch.squeeze(lp, d)
        if self._multiplier:
            lp = lp * self._multiplier
        return lp
