"""StochasticTensor: a named node = (BayesianNet, Distribution, n_samples, reduction kwargs).
Interface of zhusuan/framework/stochastic_tensor.py:5-181 of the reference."""
import contextlib
import contextvars

import torch

from .. import _hip
from .._shapes import broadcast_shapes

__all__ = ['StochasticTensor', 'LazyDraw', 'skip_discarded_draws']

# ---------------------------------------------------------------------------------------------------------------------
# The reference draws every latent TWICE per objective evaluation: once when the variational net creates the node
# (bn.py:158 / :216 return ``node.tensor``, a fresh sample), once more when the objective re-reads ``node.tensor``
# (elbo.py:122, importance_weighted_objective.py:85) -- and only the second draw is used.  The first one is kept by
# default (user code may read the value a node factory returns).  Inside ``zhusuan.skip_discarded_draws()`` an objective
# runs its variational net with node creation DEFERRED: the factories return a ``LazyDraw`` that samples only if
# somebody actually touches it, so a net that ignores the returned values (all three example callers) draws each latent
# once per step.
# ---------------------------------------------------------------------------------------------------------------------
# Both switches are CONTEXT-LOCAL (contextvars): another thread, or another asyncio task, evaluating its own objective is not
# affected by a ``with zhusuan.skip_discarded_draws():`` here (round 3 kept them in module globals).
_skip_discarded = contextvars.ContextVar("zhusuan_skip_discarded_draws", default=False)
_deferring = contextvars.ContextVar("zhusuan_deferring_node_values", default=False)


@contextlib.contextmanager
def skip_discarded_draws(enabled=True):
    """Objectives evaluated inside this context do not execute the draw that the reference discards (default: they do).

    Contract of the deferred values: inside the context a node factory of the VARIATIONAL net returns a ``LazyDraw`` instead
    of a tensor.  Touching it while the net's ``forward`` runs (any torch function, arithmetic, indexing, an attribute) draws
    the sample at that point -- exactly the value the reference's factory would have returned.  A handle that is still
    untouched when the objective re-reads the node (elbo.py:122) is EXPIRED: using it afterwards raises ``RuntimeError``
    instead of drawing out of order (the reference would have handed back the discarded first draw, which no longer exists)."""
    token = _skip_discarded.set(bool(enabled))
    try:
        yield
    finally:
        _skip_discarded.reset(token)


def skipping_discarded_draws():
    return _skip_discarded.get()


@contextlib.contextmanager
def deferred_node_values():
    """Used by the objectives around their call of the variational net (only while skip_discarded_draws is active)."""
    token = _deferring.set(True)
    try:
        yield
    finally:
        _deferring.reset(token)


def node_value(node):
    """What a BayesianNet node factory returns: ``node.tensor`` (bn.py:158,216), or a LazyDraw while deferring."""
    if _deferring.get() and node.name not in node.bn.observed:
        handle = LazyDraw(node)
        node.__dict__['_lazy_handle'] = handle
        return handle
    return node.tensor


def expire_deferred_value(node):
    """Called by the objectives when they re-read a node (elbo.py:122): a deferred value nobody has touched by now stands for
    the draw the reference discards, and must not be drawn later (it would consume the next call id of the stream and hand the
    caller a sample that belongs to no evaluation)."""
    handle = getattr(node, '__dict__', {}).pop('_lazy_handle', None)
    if handle is not None and handle._value is None:
        object.__setattr__(handle, "_expired", True)


def _materialize(v):
    if isinstance(v, LazyDraw):
        return v.materialize()
    if isinstance(v, (list, tuple)):
        return type(v)(_materialize(u) for u in v)
    if isinstance(v, dict):
        return dict((k, _materialize(u)) for k, u in v.items())
    return v


class LazyDraw(object):
    """The value of a freshly created node that nobody has looked at yet.  Any use -- a torch function, a tensor method
    or attribute, arithmetic, indexing -- draws the sample (once: ``node.tensor`` at that moment, exactly the value the
    reference's factory would have returned) and carries on with the tensor."""

    def __init__(self, node):
        object.__setattr__(self, "_node", node)
        object.__setattr__(self, "_value", None)
        object.__setattr__(self, "_expired", False)

    def materialize(self):
        if self._value is None:
            if self._expired:
                raise RuntimeError(
                    "zhusuan.skip_discarded_draws: the value that the node factory of '%s' returned was not used before the "
                    "objective drew the node again, so it stands for the draw the reference discards and was never made. Use it "
                    "inside the variational net's forward(), or evaluate the objective outside skip_discarded_draws()." % self._node.name)
            object.__setattr__(self, "_value", self._node.tensor)
        return self._value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return func(*_materialize(args), **_materialize(kwargs or {}))

    def __getattr__(self, name):
        return getattr(self.materialize(), name)

    def __repr__(self):
        return "LazyDraw(%r%s)" % (self._node.name, (", expired" if self._expired else "") if self._value is None else ", drawn")

    def __len__(self):
        return len(self.materialize())

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __bool__(self):
        return bool(self.materialize())

    def __float__(self):
        return float(self.materialize())

    def __neg__(self):
        return -self.materialize()


def _binary(name):
    def op(self, other):
        return getattr(self.materialize(), name)(_materialize(other))
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "matmul", "pow", "radd", "rsub", "rmul", "rtruediv", "rmatmul", "rpow",
           "lt", "le", "gt", "ge", "eq", "ne"):
    setattr(LazyDraw, "__%s__" % _n, _binary("__%s__" % _n))


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
        cached = self.__dict__.get('_plan_cache')          # (asked two or three times per node and step for the same value)
        if cached is not None and cached[0] is x:
            return cached[1]
        plan = self._reduction_plan_of(x, dist, g)
        self.__dict__['_plan_cache'] = (x, plan)
        return plan

    def _reduction_plan_of(self, x, dist, g):
        full = tuple(broadcast_shapes(tuple(torch.as_tensor(x).shape), tuple(dist.batch_shape)))
        nd = len(full) - g  # ndim of dist.log_prob(x)
        mean_dims = _norm_dims(self._reduce_mean_dims, nd)
        sum_dims = _norm_dims(self._reduce_sum_dims, nd)
        extra = 0
        while nd - 1 - extra >= 0 and (nd - 1 - extra) in sum_dims and (nd - 1 - extra) not in mean_dims:
            extra += 1
        return full, nd, mean_dims, sum_dims, extra

    def _scalar_coef(self, sample=None):
        """When the node's reductions collapse EVERY axis (the VAE / BNN callers), ``log_prob()`` is
        ``coef * (sum of all element-wise log-probs)``: returns ``(coef, n_fold)`` -- coef = multiplier / prod(sizes of the
        mean axes), since means and sums over distinct axes commute; n_fold = trailing axes a row-sum kernel may fold --
        or None when some axis survives.  Launches nothing."""
        full, nd, mean_dims, sum_dims, extra = self._reduction_plan(sample)
        if nd == 0 or set(mean_dims) | set(sum_dims) != set(range(nd)) or set(mean_dims) & set(sum_dims):
            return None
        coef = 1.0
        for d in mean_dims:
            coef /= float(full[d])
        if self._multiplier:
            coef *= float(self._multiplier)
        return coef, self._dist.group_ndims + extra

    def _scalar_term(self, sample=None, rows_only=False):
        """The node as one term of the one-launch scalar objective (LJ1, ``_ops.LogJointScalar``):
        ``(family, (x, a, b), (px, pa, pb), n, coef)``.  Families with a term form (Normal, Bernoulli) are evaluated INSIDE
        that launch; a fused log-density that the sampling kernel has already produced, and every other family's
        ``_log_prob_sum`` result, enter as ready-made rows (``LJ_ROWS``; with ``rows_only`` every node does: its own tuned
        log-prob kernel runs, LJ1 only adds the rows up).  None when the node does not reduce to a scalar."""
        sc = self._scalar_coef(sample)
        if sc is None:
            return None
        coef, n_fold = sc
        term = None
        if not rows_only and hasattr(self._dist, '_lj_term'):
            term = self._dist._lj_term(sample, n_fold)
        if term is None:
            rows = self._dist._log_prob_sum(sample, n_fold)
            from .._ops import _dense_flat
            flat = _dense_flat(rows)
            return _hip.LJ_ROWS, (flat, None, None), (flat.numel(), 1, 1), flat.numel(), coef
        fam, operands, periods, n = term
        return fam, operands, periods, n, coef

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
