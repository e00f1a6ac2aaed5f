"""Importance-weighted objective (IWAE / VIMCO).  Interface of
zhusuan/variational/importance_weighted_objective.py:28-191 of the reference."""
import torch
import torch.nn as nn

from ..framework.stochastic_tensor import StochasticTensor
from .. import _ops
from .elbo import latent_value, draw_latents, run_variational
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
        """importance_weighted_objective.py:79-100.  The generator's log-joint is handed to the kernel as (up to) two
        terms, so that its last addition (e.g. log p(x|z) + log p(z)), the subtraction of log q, the K-particle
        reductions and the batch mean are ONE launch; the value is rounded exactly like the reference's separate ops."""
        run_variational(self.variational, observed)
        nodes_q = self.variational.nodes
        if self.estimator == "vimco":
            _v_inputs = {}
            for k, v in nodes_q.items():          # (the reference draws the node, then checks it: :85-89)
                _v_inputs[k] = latent_value(v)
                if isinstance(v, StochasticTensor) and v.dist.is_reparameterized:
                    raise ValueError("with vimco estimator, the is_reparameterized must be false")
        else:
            _v_inputs = draw_latents(nodes_q)
        _observed = {**_v_inputs, **observed}
        nodes_p = self.generator(_observed).nodes
        cls = type(self)
        if (cls.log_joint is not ImportanceWeightedObjective.log_joint or cls.sgvb is not ImportanceWeightedObjective.sgvb
                or cls.vimco is not ImportanceWeightedObjective.vimco):
            # a subclass overrides one of the reference's hooks: keep calling them like the reference does (:97-100)
            logpxz, logqz = self.log_joint(nodes_p), self.log_joint(nodes_q)
            return self.sgvb(logpxz, logqz, reduce_mean) if self.estimator == 'sgvb' else self.vimco(logpxz, logqz, reduce_mean)
        terms_p = [nodes_p[n].log_prob() for n in nodes_p.keys()]
        logqz = self.log_joint(nodes_q)
        head = None
        for t in terms_p[:-1]:                      # left-to-right, as log_joint sums them
            head = t if head is None else head + t
        if head is None:
            return self._objective(terms_p[0], None, logqz, reduce_mean)
        return self._objective(head, terms_p[-1], logqz, reduce_mean)

    def _rows(self, tensors, vimco=False):
        """The given log-joint tensors as K-fastest [B, K] matrices plus K and the shape of the non-particle axes.
        Tensors produced by the log-prob kernels already have this layout, so no copy happens."""
        tensors = [torch.as_tensor(t) for t in tensors]
        dev = tensors[0].device
        tensors = [t if t.device == dev else t.to(dev) for t in tensors]
        shape = broadcast_shapes(*[tuple(t.shape) for t in tensors])
        nd = len(shape)
        if nd == 0:
            raise ValueError(_ERR_VIMCO)
        if not -nd <= self._axis < nd:
            if vimco:
                raise ValueError(_ERR_VIMCO)       # _shape[self._axis] fails inside the reference's size check (:154-162)
            raise IndexError("Dimension out of range (expected to be in range of [%d, %d], but got %d)"
                             % (-nd, nd - 1, self._axis))
        if vimco:
            if shape[self._axis] < 2:
                raise ValueError(_ERR_VIMCO)
            # The reference's VIMCO (:164-186) works for a 1-D log_w and for a 2-D one whose particle axis is 0; every
            # other layout fails there, and fails here with the same exception type: a negative axis in F.one_hot (:173),
            # three or more axes in torch.transpose(x, *perm) (:183), [B, K] with axis=1 in the broadcast of :186 (its
            # permutation is the identity, so the diagonal terms no longer line up; for a SQUARE log_w the reference
            # returns a meaningless number instead of raising -- refused here as well).
            if self._axis < 0:
                raise RuntimeError("Class values must be non-negative.")
            if nd >= 3:
                raise TypeError("transpose() received an invalid combination of arguments - got (Tensor, %s), but expected "
                                "(Tensor input, int dim0, int dim1): VIMCO supports a 1-D or [K, B] log-weight tensor "
                                "(importance_weighted_objective.py:176-183)" % ", ".join(["Tensor"] * nd))
            if nd == 2 and self._axis != 0:
                raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton "
                                   "dimension 2: VIMCO needs the particle axis first (axis=0)" % (shape[1], shape[0]))
        axis = self._axis % nd
        K = shape[axis]
        rest = tuple(s for i, s in enumerate(shape) if i != axis)

        def rows(t):
            t = t.expand(shape) if tuple(t.shape) != tuple(shape) else t
            return t.movedim(axis, -1).reshape(-1, K)
        return [rows(t) for t in tensors], K, rest

    def _objective(self, logp_a, logp_b, logqz, reduce_mean, estimator=None):
        vimco = (estimator or self.estimator) == 'vimco'
        try:
            mats, K, rest = self._rows([logp_a, logqz] if logp_b is None else [logp_a, logp_b, logqz], vimco)
        except ZeroDivisionError:
            raise ValueError(_ERR_VIMCO)
        if vimco and K < 2:
            raise ValueError(_ERR_VIMCO)
        a2, q2 = mats[0], mats[-1]
        b2 = mats[1] if logp_b is not None else None
        want_mean = True if vimco else bool(reduce_mean)      # vimco always returns the batch mean (:191)
        cost, bound_b = _ops.IWObjective.apply(a2, b2, q2, _ops.ZS_IW_VIMCO if vimco else _ops.ZS_IW_SGVB, want_mean)
        self.last_iw_bound = bound_b.reshape(rest)
        return cost if want_mean else cost.reshape(rest)

    def sgvb(self, logpxz, logqz, reduce_mean=True):
        """importance_weighted_objective.py:102-132 with compute_iw_term (:16-25), one kernel."""
        return self._objective(logpxz, None, logqz, reduce_mean, 'sgvb')

    def vimco(self, logpxz, logqz, reduce_mean=True):
        """importance_weighted_objective.py:134-191.  Always returns the batch mean, like the reference."""
        return self._objective(logpxz, None, logqz, reduce_mean, 'vimco')
