"""Univariate Uniform on [low, high).  API of zhusuan/distributions/uniform.py:9-88 of the reference; kernels U1 / U2
of include/zs_hip.h."""
import torch

from .base import Distribution
from .utils import assert_same_log_float_dtype, check_broadcast
from .. import _hip, _ops, _rng
from .._shapes import broadcast_shapes, value_shape

__all__ = ['Uniform']


def _validating(device):
    """The reference evaluates log-probs through torch.distributions.Uniform, whose argument / support checks
    (ValueError) follow torch's global switch.  They read the device, so they are skipped during hipGraph capture."""
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        return False
    return bool(torch.distributions.Distribution._validate_args)


class Uniform(Distribution):
    """
    :param low / high: float tensors, lower (inclusive) and upper (exclusive) range.
    :param is_reparameterized: True: u * (high - low) + low with u ~ U(0,1) of LOW's shape; False: the reference
        draws from Uniform(low, high) and then applies the same affine map AGAIN (uniform.py:63-64,70) -- kept.
    ``sample_cache`` holds the value BEFORE the final affine map (uniform.py:69), as in the reference.
    """

    def __init__(self,
                 low,
                 high,
                 dtype=None,
                 is_continuous=True,
                 is_reparameterized=True,
                 group_ndims=0,
                 device=None,
                 **kwargs):
        device = _hip.resolve_device(device, low, high)
        self._low = torch.as_tensor(low, dtype=dtype).to(device)
        self._high = torch.as_tensor(high, dtype=dtype).to(device)
        check_broadcast(self._low, self._high)
        dtype = assert_same_log_float_dtype([(self._low, "Uniform.low"), (self._high, "Uniform.high")])
        super(Uniform, self).__init__(dtype=dtype,
                                      is_continuous=is_continuous,
                                      is_reparameterized=is_reparameterized,
                                      group_ndims=group_ndims,
                                      device=device,
                                      **kwargs)

    @property
    def low(self):
        """Lower range (inclusive)."""
        return self._low

    @property
    def high(self):
        """Upper range (exclusive)."""
        return self._high

    def _batch_shape(self):
        return torch.Size(broadcast_shapes(self._low.shape, self._high.shape))

    def _check_args(self):
        if _validating(self._low.device) and not bool(torch.lt(self._low, self._high).all()):
            raise ValueError("Expected parameter low of distribution Uniform to satisfy the constraint "
                             "LessThan(upper_bound=high)")

    def _sample(self, n_samples=1, uniform=None, **kwargs):
        """uniform.py:51-70."""
        K = int(n_samples)
        lead = (K,) if K > 1 else ()
        low, high = self._low, self._high
        bshape = tuple(self._batch_shape())
        full = lead + bshape
        reparam = bool(self._is_reparameterized)
        if not reparam:
            self._check_args()               # torch.distributions.Uniform(low, high) is built at uniform.py:64
        u_shape = lead + (tuple(low.shape) if reparam else bshape)
        u = uniform
        if u is None:
            u = _rng.pop_injected(u_shape, low.device, low.dtype, kind="rand")
        else:
            u = torch.as_tensor(u, dtype=low.dtype).to(low.device)
            if tuple(u.shape) != u_shape:
                raise RuntimeError("uniform draw has shape %s, expected %s" % (tuple(u.shape), u_shape))
        seed = call = 0
        rng_state = None
        if u_shape == full:
            if u is None:
                seed, call, rng_state = _rng.next_call(low.device)
            else:
                u = u.contiguous()
        else:                                # draw shared along the axes where only `high` broadcasts
            if u is None:
                s, c, rs = _rng.next_call(low.device)
                u = _ops.philox_uniform(u_shape, low.device, s, c, rs, low.dtype)
            pad = (1,) * (len(bshape) - low.dim())
            u = u.reshape(lead + pad + tuple(low.shape)).expand(full).contiguous()
        pl, Pl = _ops.periodic_operand(low, full)
        ph, Ph = _ops.periodic_operand(high, full)
        out, cache = _ops.UniformSample.apply(pl, ph, u, seed, call, rng_state, full, (Pl, Ph), reparam)
        self.sample_cache = cache
        return out

    def _log_prob_sum(self, given=None, n_fold=0):
        """uniform.py:72-85 (+ trailing sum over `n_fold` axes)."""
        x = self.sample_cache if given is None else given
        if x is None:
            raise RuntimeError("Uniform.log_prob(None) needs a cached sample: call sample() first")
        x = torch.as_tensor(x, dtype=self._dtype).to(self._low.device)
        self._check_args()
        try:
            full = value_shape(x.shape, self._low.dim(), self._low.shape, self._high.shape)
        except RuntimeError as e:
            # torch.distributions.Uniform(_low, _high).log_prob validates the value's shape first (uniform.py:82)
            raise ValueError("Value is not broadcastable with batch_shape+event_shape: %s" % e)
        if n_fold > len(full):
            raise ValueError("cannot sum %d trailing axes of a result of shape %s" % (n_fold, full))
        if _validating(x.device) and not bool((torch.ge(x, self._low) & torch.le(x, self._high)).all()):
            raise ValueError("Expected value argument to be within the support "
                             "(Interval(lower_bound=low, upper_bound=high)) of the distribution Uniform")
        px, Px = _ops.periodic_operand(x, full)
        pl, Pl = _ops.periodic_operand(self._low, full)
        ph, Ph = _ops.periodic_operand(self._high, full)
        return _ops.UniformLogProb.apply(px, pl, ph, full, n_fold, (Px, Pl, Ph), True)
