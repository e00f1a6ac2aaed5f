"""Importance-weighted objective (IWAE / VIMCO).  Interface of
zhusuan/variational/importance_weighted_objective.py:28-191 of the reference."""
import torch
import torch.nn as nn

from ..framework.stochastic_tensor import StochasticTensor
from .. import _ops
from .elbo import latent_value, draw_latents, run_variational
from .._shapes import broadcast_shapes
from ..distributions.bernoulli import Bernoulli
from ..distributions.normal import Normal

__all__ = ['ImportanceWeightedObjective']

_ERR_VIMCO = ("VIMCO is a multi-sample gradient estimator, size along "
              "`axis` in the objective should be larger than 1.")


from ..utils import note_path as _note_path      # noqa: E402

_PATH_FUSED = "IW1: the generator side of the objective in one launch each way (zs_bernoulli_iw_objective)"
_PATH_NODES = "per-node kernels (K3 / K2 per node) + the objective in one launch (K4b): three or more launches each way"

_LOGQ = object()      # key of log q among the log-probabilities a fused attempt hands to the per-node path (cannot be a node's name)


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
        self.last_path = None          # which kernels the last evaluation ran on, and why (zhusuan.explain)

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
            _note_path(self, _PATH_NODES, "a subclass overrides log_joint / sgvb / vimco: its hooks are called as the reference calls them")
            return self.sgvb(logpxz, logqz, reduce_mean) if self.estimator == 'sgvb' else self.vimco(logpxz, logqz, reduce_mean)
        done = {}           # log-probabilities the fused attempt has already evaluated (a layout rejected late used to
        why = []
        fused = self._generator_side_in_one_launch(nodes_p, nodes_q, reduce_mean, done, why)  # evaluate them a second time below)
        if fused is not None:
            _note_path(self, _PATH_FUSED)
            return fused
        _note_path(self, _PATH_NODES, why[0] if why else "the model's layout is not the one the fused kernel covers")
        terms_p = [done[n] if n in done else nodes_p[n].log_prob() for n in nodes_p.keys()]
        logqz = done[_LOGQ] if _LOGQ in done else self.log_joint(nodes_q)
        head = None
        for t in terms_p[:-1]:                      # left-to-right, as log_joint sums them
            head = t if head is None else head + t
        if head is None:
            return self._objective(terms_p[0], None, logqz, reduce_mean)
        return self._objective(head, terms_p[-1], logqz, reduce_mean)

    def _generator_side_in_one_launch(self, nodes_p, nodes_q, reduce_mean, done, why=None):
        """IW1 (``_ops.BernoulliIWObjective``): when the generator's LAST node is a Bernoulli likelihood over [K, B, X] whose
        log-probability reduces to [K, B] -- the IWAE caller, examples/variational_autoencoder/iwae.py:49-81 -- its row sums,
        the log-density of the latent under a Normal prior node, the sum of the generator's terms, the subtraction of log q,
        the K-particle reductions and the batch mean are ONE launch instead of the per-node loop (:66-100) + K4b; the backward
        is one call as well, which also takes over the gradient of a non-reparameterised Normal q node.  Returns None when
        the layout is anything else (the per-node path then runs; ``why`` receives the reason, for ``zhusuan.explain``)."""
        why = [] if why is None else why
        if self._axis != 0 or not nodes_p or not nodes_q:
            why.append("the particle axis is not 0 (or a net has no nodes)")
            return None
        vimco = self.estimator == 'vimco'
        names = list(nodes_p.keys())
        last = nodes_p[names[-1]]
        plan = _bernoulli_rows_plan(last)
        if plan is None:
            why.append("the generator's last node is not a Bernoulli likelihood over [K, B, X] whose log-probability is a plain "
                       "sum over X to [K, B] (no mean dims, no multiplier, observation [B, X] or [K, B, X])")
            return None
        par, x, Px, from_logits, (K, B, X) = plan
        if vimco and K < 2:
            return None
        reason = _ops.iw1_unsupported_reason(K, B, X, par.dtype, par, x)
        if reason is not None:
            why.append(reason)
            return None
        # the other generator nodes: ONE Normal node of the latent value becomes a term of the launch; anything else enters as
        # ready-made rows from its own kernel (added left to right like log_joint, :66-77)
        z = pmu = psigma = None
        Pm = Ps = 1
        p_ls = False
        rows_a = None
        others = names[:-1]
        if others:
            term = _normal_rows_plan(nodes_p[others[-1]], K, B)
            if term is not None:
                z, pmu, psigma, Pm, Ps, p_ls = term
                others = others[:-1]
            for n in others:
                lp = done[n] = nodes_p[n].log_prob()
                if tuple(lp.shape) != (K, B):
                    why.append("generator node %r has a log-probability of shape %s, not [K, B]" % (n, tuple(lp.shape)))
                    return None
                rows_a = lp if rows_a is None else rows_a + lp
        logqz = done[_LOGQ] = self.log_joint(nodes_q)
        if not isinstance(logqz, torch.Tensor) or tuple(logqz.shape) != (K, B) or logqz.dtype != par.dtype or logqz.device != par.device:
            why.append("log q is not a [K, B] tensor of the likelihood's dtype and device")
            return None
        if rows_a is not None and (rows_a.dtype != par.dtype or rows_a.device != par.device):
            why.append("the generator's nodes differ in dtype or device")
            return None
        # a single non-reparameterised Normal q node whose fused log-density is log q: IW1's backward forms its parameter
        # gradients itself (log q then enters detached; the sampler's own backward is not needed)
        qmu = qsigma = qz = None
        q_ls = False
        if len(nodes_q) == 1:
            fold = _foldable_q_node(next(iter(nodes_q.values())), logqz, K, B)
            if fold is not None:
                qmu, qsigma, qz, q_ls = fold
                logqz = logqz.detach()
        want_mean = True if vimco else bool(reduce_mean)          # vimco always returns the batch mean (:191)
        meta = (from_logits, Px, Pm, Ps, p_ls, _ops.ZS_IW_VIMCO if vimco else _ops.ZS_IW_SGVB, want_mean, q_ls)
        cost, bound_b = _ops.BernoulliIWObjective.apply(par, x, z, pmu, psigma, None if rows_a is None else rows_a.t(), logqz.t(),
                                                        qmu, qsigma, qz, meta)
        self.last_iw_bound = bound_b
        return cost

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


# ---------------------------------------------------------------------------------------------------------------------
# Layout checks of the one-launch generator side (IW1).  Each returns None when the node is not of the simple layout the
# fused kernel covers; nothing is launched by asking.
# ---------------------------------------------------------------------------------------------------------------------
def _plain_rows_node(node, cls):
    """The node's distribution when it is exactly a `cls` node whose log_prob is a plain trailing sum (no mean, no multiplier)."""
    dist = getattr(node, 'dist', None)
    if type(dist) is not cls or not isinstance(node, StochasticTensor):
        return None
    if node._multiplier or node._reduce_mean_dims:
        return None
    return dist


def _bernoulli_rows_plan(node):
    """(parameter [K, B, X], observation, its period, from_logits, (K, B, X)) of a Bernoulli node whose log_prob is [K, B]."""
    dist = _plain_rows_node(node, Bernoulli)
    if dist is None:
        return None
    par = dist._param()
    x = dist.sample_cache
    if not isinstance(x, torch.Tensor) or par.dim() != 3 or par.dtype not in (torch.float32, torch.float64):
        return None
    K, B, X = par.shape
    try:
        full, nd, mean_dims, sum_dims, extra = node._reduction_plan()
    except (RuntimeError, IndexError):
        return None
    if tuple(full) != (K, B, X) or nd - extra != 2 or dist.group_ndims + extra != 1 or mean_dims or any(d < nd - extra for d in sum_dims):
        return None
    if K * B * X == 0 or x.dtype != par.dtype or x.device != par.device:
        return None
    if tuple(x.shape) == (B, X):
        Px = B * X
    elif tuple(x.shape) == (K, B, X):
        Px = K * B * X
    else:
        return None
    return par.contiguous(), x.contiguous(), Px, dist._from_logits, (K, B, X)


def _normal_rows_plan(node, K, B):
    """(value [K, B, Dz], mean, scale operand, their periods, scale_is_logstd) of a Normal node of a given [K, B, Dz] value
    whose log_prob is [K, B] and whose parameters are [B, Dz] (repeated over the particles) or scalars."""
    dist = _plain_rows_node(node, Normal)
    if dist is None:
        return None
    z = dist.sample_cache
    if not isinstance(z, torch.Tensor) or z.dim() != 3 or tuple(z.shape[:2]) != (K, B):
        return None
    if dist._fused is not None and dist._fused[0] is z:
        return None                      # the node drew this value itself: its fused log-density exists already (rows)
    Dz = z.shape[2]
    try:
        full, nd, mean_dims, sum_dims, extra = node._reduction_plan()
    except (RuntimeError, IndexError):
        return None
    if tuple(full) != (K, B, Dz) or nd - extra != 2 or dist.group_ndims + extra != 1 or mean_dims or any(d < nd - extra for d in sum_dims):
        return None
    mean, scale = dist._mean, dist._scale_operand()
    ops = []
    for t in (mean, scale):
        if t.dtype != z.dtype or t.device != z.device:
            return None
        if t.numel() == 1 and B * Dz != 1:
            ops.append((t.reshape(1), 1))
        elif tuple(t.shape) == (B, Dz) or tuple(t.shape) == (1, B, Dz):
            ops.append((t.contiguous(), B * Dz))
        else:
            return None
    z = z.contiguous()
    if Dz == 0 or not _ops.iw1_term_supported(Dz, z.dtype, z, ops[0][0], ops[1][0]):
        return None
    return z, ops[0][0], ops[1][0], ops[0][1], ops[1][1], dist._logstd_given is not None


def _foldable_q_node(node, logqz, K, B):
    """(mean, scale operand, draw, scale_is_logstd) when `logqz` IS the fused log-density of this non-reparameterised Normal
    node's current draw z [K, B, Dq] with parameters [B, Dq]: then d(-objective)/d(mean, scale) is a K-summed function of
    (z, mean, scale) and the objective's coefficients, which IW1's backward evaluates itself (normal.py:102,112-116)."""
    dist = _plain_rows_node(node, Normal)
    if dist is None or dist.is_reparameterized or dist._fused is None:
        return None
    z, lp, n_fold = dist._fused
    if lp is not logqz or z is not dist.sample_cache or z.dim() != 3 or tuple(z.shape[:2]) != (K, B) or n_fold != 1:
        return None
    mean, scale = dist._mean, dist._scale_operand()
    if tuple(mean.shape) != tuple(z.shape[1:]) or tuple(scale.shape) != tuple(z.shape[1:]):
        return None
    if not (mean.requires_grad or scale.requires_grad):
        return None
    return mean.contiguous(), scale.contiguous(), z.detach().contiguous(), dist._logstd_given is not None
