"""Univariate Logistic.  API of zhusuan/distributions/logistic.py:10-86 of the reference; the sample and its
log-density come from the fused HIP kernel L1, the density of a given value from L2 (include/zs_hip.h)."""
import torch

from .base import Distribution
from .utils import assert_same_log_float_dtype, check_broadcast
from .. import _hip, _ops, _rng
from .._shapes import broadcast_shapes, value_shape

__all__ = ['Logistic']


def _capturing(device):
    return device.type == "cuda" and torch.cuda.is_current_stream_capturing()


class Logistic(Distribution):
    """
    :param loc: float tensor, location.
    :param scale: float tensor, > 0 (ValueError otherwise, logistic.py:30-31; the check reads the device and is
        skipped while a hipGraph is being captured).
    Always reparameterised (logistic.py:36): z = loc + scale * (log u - log(1 - u)).
    """

    def __init__(self,
                 loc,
                 scale,
                 dtype=None,
                 is_continuous=True,
                 group_ndims=0,
                 device=None,
                 **kwargs):
        device = _hip.resolve_device(device, loc, scale)
        self._loc = torch.as_tensor(loc, dtype=dtype).to(device)
        self._scale = torch.as_tensor(scale, dtype=dtype).to(device)
        if not _capturing(self._scale.device) and bool(torch.less_equal(self._scale, 0.).any()):
            raise ValueError("scale less than zero")
        check_broadcast(self._loc, self._scale)
        dtype = assert_same_log_float_dtype([(self._loc, "Logistic.loc"), (self._scale, "Logistic.scale")])
        super(Logistic, self).__init__(dtype=dtype,
                                       is_continuous=is_continuous,
                                       is_reparameterized=True,
                                       group_ndims=group_ndims,
                                       device=device,
                                       **kwargs)
        self._fused = None  # (sample tensor, its row-summed log-density, n_fold)

    @property
    def loc(self):
        return self._loc

    @property
    def scale(self):
        return self._scale

    def _batch_shape(self):
        return torch.Size(broadcast_shapes(self._loc.shape, self._scale.shape))

    def _sample(self, n_samples=1, uniform=None, **kwargs):
        """logistic.py:52-67.  The U(0,1) draw has LOC's shape (``[K] + loc.shape``, :54-55,61,64).  `uniform`
        (or zhusuan.inject_epsilon) supplies it explicitly; otherwise it comes from the in-kernel Philox stream."""
        K = int(n_samples)
        has_k = K > 1
        loc, scale = self._loc, self._scale
        bshape = tuple(self._batch_shape())
        lead = (K,) if has_k else ()
        u_shape = lead + tuple(loc.shape)
        u = uniform
        if u is None:
            u = _rng.pop_injected(u_shape, loc.device, loc.dtype, kind="uniform_init")
        else:
            u = torch.as_tensor(u, dtype=loc.dtype).to(loc.device)
            if tuple(u.shape) != u_shape:
                raise RuntimeError("uniform draw has shape %s, expected %s" % (tuple(u.shape), u_shape))
        seed = call = 0
        rng_state = None
        if tuple(loc.shape) == tuple(scale.shape):
            lo, sc = loc.contiguous(), scale.contiguous()
            if u is None:
                seed, call, rng_state = _rng.next_call(loc.device)
            else:
                u = u.contiguous()
        else:
            if u is None:
                s, c, rs = _rng.next_call(loc.device)
                u = _ops.philox_uniform(u_shape, loc.device, s, c, rs, loc.dtype)
            pad = (1,) * (len(bshape) - loc.dim())
            u = u.reshape(lead + pad + tuple(loc.shape)).expand(lead + bshape).contiguous()
            lo = loc.expand(bshape).contiguous()
            sc = scale.expand(bshape).contiguous()
        n_fold = min(max(1, self._group_ndims), len(bshape))
        z, lp = _ops.LogisticSampleLogProb.apply(lo, sc, u, seed, call, rng_state, K if has_k else 1, has_k, n_fold, True)
        self.sample_cache = z
        self._fused = (z, lp, n_fold)
        return z

    def _log_prob_sum(self, given=None, n_fold=0):
        """logistic.py:69-83 (+ trailing sum over `n_fold` axes)."""
        x = self.sample_cache if given is None else given
        if x is None:
            raise RuntimeError("Logistic.log_prob(None) needs a cached sample: call sample() first")
        if self._fused is not None and self._fused[0] is x and self._fused[2] == n_fold:
            return self._fused[1]
        x = torch.as_tensor(x, dtype=self._dtype).to(self._loc.device)
        full = value_shape(x.shape, self._loc.dim(), self._loc.shape, self._scale.shape)
        if n_fold > len(full):
            raise ValueError("cannot sum %d trailing axes of a result of shape %s" % (n_fold, full))
        px, Px = _ops.periodic_operand(x, full)
        pm, Pm = _ops.periodic_operand(self._loc, full)
        ps, Ps = _ops.periodic_operand(self._scale, full)
        return _ops.LogisticLogProb.apply(px, pm, ps, full, n_fold, (Px, Pm, Ps), True)
