"""Constructor checks of the distributions: parameter dtypes and broadcastability.

Behavioural contract taken from zhusuan/distributions/utils.py:12-71 of the reference: the exception
types and message texts below are what its tests match on (test/distributions/test_normal.py:31,
test_bernoulli.py:29-33).
"""
import torch

from .._shapes import broadcast_shapes

# dtypes on which log / exp are defined for the kernels (fp32 tuned, fp64 plain)
log_floating_dtypes = (torch.float32, torch.float64)


def _dtype_error(name, dtype, allowed):
    if len(allowed) == 1:
        return TypeError('{}({}) must have dtype {}.'.format(name, dtype, allowed[0]))
    return TypeError('{}({}) must have a dtype in {}.'.format(name, dtype, allowed))


def assert_same_dtype_in(tensors_with_name, dtypes=None):
    """Return the common dtype of the (tensor, name) pairs; TypeError if one is outside `dtypes`
    or if two of them differ."""
    pairs = list(tensors_with_name)
    if not pairs:
        return None
    if dtypes:
        for tensor, name in pairs:
            if tensor.dtype not in dtypes:
                raise _dtype_error(name, tensor.dtype, dtypes)
    ref_tensor, ref_name = pairs[0]
    for tensor, name in pairs[1:]:
        if tensor.dtype != ref_tensor.dtype:
            raise TypeError('{}({}) must have the same dtype as {}({}).'.format(
                name, tensor.dtype, ref_name, ref_tensor.dtype))
    return ref_tensor.dtype


def assert_same_log_float_dtype(tensors_with_name):
    """All tensors float32, or all float64."""
    return assert_same_dtype_in(tensors_with_name, log_floating_dtypes)


def check_broadcast(mean, std):
    """RuntimeError when the two parameter shapes do not broadcast.  (The reference finds out by evaluating
    ``mean + std``; only the shapes matter, so no kernel is launched here.)"""
    broadcast_shapes(tuple(mean.shape), tuple(std.shape))
