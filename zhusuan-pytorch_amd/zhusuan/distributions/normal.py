"""Univariate Normal.  API of zhusuan/distributions/normal.py:12-129 of the reference; the sample and its
log-density come from the fused HIP kernel K1, the density of a given value from K2 (include/zs_hip.h)."""
import torch

from .base import Distribution
from .utils import assert_same_log_float_dtype
from .. import _hip, _ops, _rng
from .._shapes import broadcast_shapes, value_shape

__all__ = ['Normal']


def _as_param(v, dtype, device):
    """``torch.as_tensor(v, dtype=dtype).to(device)`` (normal.py:50,56,58), without the two calls when `v` is already that."""
    if isinstance(v, torch.Tensor) and (dtype is None or v.dtype == dtype) and v.device == device:
        return v
    return torch.as_tensor(v, dtype=dtype).to(device)


class Normal(Distribution):
    _nonreparam_draw_has_zero_grad = True      # see Distribution
    """
    :param mean: float tensor (or anything ``torch.as_tensor`` accepts); broadcastable against std.
    :param std / logstd: exactly one of them (ValueError otherwise, normal.py:51-54).
    :param is_reparameterized: True: z = mean + std * eps carries gradients to (mean, std)
        (normal.py:104-105); False: the draw is detached like ``torch.normal(mean, std)`` (:102).
    :param group_ndims: trailing batch axes summed into one event (base.py:175-176).
    :param device: where the parameters are moved (normal.py:50).  Default: the device of the first
        tensor parameter, else the current GPU -- the reference defaults to the CPU, for which this
        build has no kernels.
    """

    def __init__(self,
                 mean=0.,
                 std=None,
                 logstd=None,
                 dtype=None,
                 is_continuous=True,
                 is_reparameterized=True,
                 group_ndims=0,
                 device=None,
                 **kwargs):
        device = _hip.resolve_device(device, mean, std, logstd)
        self._mean = _as_param(mean, dtype, device)
        if (logstd is None) == (std is None):
            raise ValueError(
                "Either `std` or `logstd` should be passed. It is not allowed "
                "that both are specified or both are not.")
        elif std is None:
            # normal.py:56 computes std = exp(logstd) here.  The kernels take log(sigma) directly (sigma_is_logstd of
            # include/zs_hip.h) and form exp() in registers, so constructing the node launches nothing; `std` is
            # materialised only if somebody reads the property.
            self._logstd_given = _as_param(logstd, dtype, device)
            self._std_cache = None
        else:
            self._logstd_given = None
            self._std_cache = _as_param(std, dtype, device)
        scale = self._scale_operand()
        # (the broadcast check IS the batch shape: kept, the eager path asks for it several times per step)
        self._bshape = torch.Size(broadcast_shapes(self._mean.shape, scale.shape))
        dtype = assert_same_log_float_dtype([(self._mean, "Normal.mean"), (scale, "Normal.std")])
        super(Normal, self).__init__(dtype=dtype,
                                     is_continuous=is_continuous,
                                     is_reparameterized=is_reparameterized,
                                     group_ndims=group_ndims,
                                     device=device,
                                     **kwargs)
        self._fused = None  # (sample tensor, its row-summed log-density, n_fold)

    @property
    def mean(self):
        return self._mean

    def _scale_operand(self):
        """The tensor handed to the kernels as `sigma`: log(sigma) for Normal(logstd=...), else sigma."""
        return self._std_cache if self._logstd_given is None else self._logstd_given

    @property
    def _std(self):
        if self._std_cache is None:
            self._std_cache = torch.exp(self._logstd_given)        # normal.py:56
        return self._std_cache

    @property
    def std(self):
        return self._std

    @property
    def logstd(self):
        return torch.log(self._std)         # normal.py:77-79 (log of exp of the argument, as in the reference)

    def _batch_shape(self):
        return self._bshape

    def _sample_plan(self, n_samples=1, epsilon=None):
        """Operands of one draw (normal.py:89-107): ``(mu, sigma, eps, K, has_k, n_fold, is_logstd, simple)``.  The
        standard-normal draw has MEAN's shape (``[K] + mean.shape``), so it is shared along axes where only std
        broadcasts.  `epsilon` (or zhusuan.inject_epsilon) supplies the draw explicitly; ``eps is None`` means the kernel
        draws from its Philox stream.  ``simple``: mean and std have one shape (no expansion was needed)."""
        K = int(n_samples)
        has_k = K > 1
        mean, std = self._mean, self._scale_operand()
        is_logstd = self._logstd_given is not None
        bshape = tuple(self._batch_shape())
        lead = (K,) if has_k else ()
        # reparameterised: torch.normal(0, 1, size=[K] + mean.shape) (normal.py:90-92,104); otherwise
        # torch.normal(mean_rep, std_rep) (normal.py:102): one independent draw per element of the BROADCAST shape
        eps_shape = lead + (tuple(mean.shape) if self._is_reparameterized else bshape)
        eps = epsilon
        if eps is None:
            eps = _rng.pop_injected(eps_shape, mean.device, mean.dtype)
        else:
            eps = torch.as_tensor(eps, dtype=mean.dtype).to(mean.device)
            if tuple(eps.shape) != eps_shape:
                raise RuntimeError("epsilon has shape %s, expected %s" % (tuple(eps.shape), eps_shape))
        simple = tuple(mean.shape) == tuple(std.shape)
        if simple:
            mu, sigma = mean.contiguous(), std.contiguous()
            if eps is not None:
                eps = eps.contiguous()
        else:
            if eps is None:
                s, c, rs = _rng.next_call(mean.device)
                eps = _ops.philox_normal(eps_shape, mean.device, s, c, rs, mean.dtype)
            if self._is_reparameterized:
                pad = (1,) * (len(bshape) - mean.dim())
                eps = eps.reshape(lead + pad + tuple(mean.shape)).expand(lead + bshape)
            eps = eps.contiguous()
            mu = mean.expand(bshape).contiguous()
            sigma = std.expand(bshape).contiguous()
        n_fold = min(max(1, self._group_ndims), len(bshape))
        return mu, sigma, eps, K, has_k, n_fold, is_logstd, simple

    def _sample(self, n_samples=1, epsilon=None):
        """normal.py:89-107 through K1: the sample and its row-summed log-density in one launch."""
        pending = self.__dict__.pop('_pending_draw', None)
        if pending is not None and epsilon is None and pending[0] == int(n_samples) and _rng.pair_draw_consumable():
            # the second of the two draws made in one launch when the node was created (below): this IS the objective's re-read
            self._adopt_draw(pending[1], pending[2], pending[3])
            return pending[1]
        mu, sigma, eps, K, has_k, n_fold, is_logstd, simple = self._sample_plan(n_samples, epsilon)
        seed = call = 0
        rng_state = None
        if (eps is None and simple and _rng.pair_draw_wanted() and mu.dtype == torch.float32 and mu.numel() > 0
                and mu.data_ptr() % 16 == 0 and sigma.data_ptr() % 16 == 0
                and _hip.lib().pair_draw_is_one_launch(K if has_k else 1, mu.numel(), _ops._prod(tuple(mu.shape)[mu.dim() - n_fold:]))):
            # an objective is running the variational net: it will draw this node again (elbo.py:122 of the reference).  Both
            # draws in ONE launch, with the call ids two launches would have used; the second waits for the re-read.
            seed, call, rng_state = _rng.next_call(mu.device)
            _, call2, _ = _rng.next_call(mu.device)
            if call2 != call + 1:
                raise RuntimeError("zhusuan: the Philox call ids of two consecutive draws are not consecutive (%d, %d)" % (call, call2))
            z, lp, z2, lp2 = _ops.NormalSampleLogProbPair.apply(mu, sigma, seed, call, rng_state, K if has_k else 1, has_k, n_fold,
                                                                bool(self._is_reparameterized), is_logstd)
            self._adopt_draw(z, lp, n_fold)
            self.__dict__['_pending_draw'] = (int(n_samples), z2, lp2, n_fold)
            return z
        if eps is None:
            seed, call, rng_state = _rng.next_call(mu.device)
        z, lp = _ops.NormalSampleLogProb.apply(mu, sigma, eps, seed, call, rng_state, K if has_k else 1, has_k, n_fold,
                                               bool(self._is_reparameterized), True, is_logstd)
        self._adopt_draw(z, lp, n_fold)
        return z

    def _adopt_draw(self, z, lp, n_fold):
        """Make `z` (with its fused log-density `lp`, summed over `n_fold` trailing axes) the node's current sample."""
        self.sample_cache = z
        self._fused = (z, lp, n_fold)

    def _lj_term(self, given, n_fold):
        """This node's contribution to a scalar log-joint as a term of the one-launch objective (LJ1):
        ``(family, (x, a, b), (px, pa, pb), n)``, or None when the fused log-density of the cached draw already exists
        (it then enters as ready-made rows)."""
        x = self.sample_cache if given is None else given
        if x is None:
            raise RuntimeError("Normal.log_prob(None) needs a cached sample: call sample() first")
        if self._fused is not None and self._fused[0] is x and self._fused[2] == n_fold:
            return None
        # a sample drawn by the multi-node sampler carries itself once more as a separate autograd output: reading it through
        # that one sends this term's gradient to its own slot of the sampler's backward (no accumulation launch)
        x = getattr(x, '_zs_grad_alias', x)
        x = torch.as_tensor(x, dtype=self._dtype).to(self._mean.device)
        full = value_shape(x.shape, self._mean.dim(), self._mean.shape, self._scale_operand().shape)
        px, Px = _ops.periodic_operand(x, full)
        pm, Pm = _ops.periodic_operand(self._mean, full)
        ps, Ps = _ops.periodic_operand(self._scale_operand(), full)
        fam = _hip.LJ_NORMAL_LOGSTD if self._logstd_given is not None else _hip.LJ_NORMAL
        return fam, (px, pm, ps), (Px, Pm, Ps), _ops._prod(full)

    def _log_prob_sum(self, given=None, n_fold=0):
        """normal.py:109-126 (+ trailing sum over `n_fold` axes)."""
        x = self.sample_cache if given is None else given
        if x is None:
            raise RuntimeError("Normal.log_prob(None) needs a cached sample: call sample() first")
        if self._fused is not None and self._fused[0] is x and self._fused[2] == n_fold:
            return self._fused[1]
        x = torch.as_tensor(x, dtype=self._dtype).to(self._mean.device)
        full = value_shape(x.shape, self._mean.dim(), self._mean.shape, self._scale_operand().shape)
        if n_fold > len(full):
            raise ValueError("cannot sum %d trailing axes of a result of shape %s" % (n_fold, full))
        px, Px = _ops.periodic_operand(x, full)
        pm, Pm = _ops.periodic_operand(self._mean, full)
        ps, Ps = _ops.periodic_operand(self._scale_operand(), full)
        return _ops.NormalLogProb.apply(px, pm, ps, full, n_fold, (Px, Pm, Ps), True, self._logstd_given is not None)
