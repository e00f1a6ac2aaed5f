"""dtype / broadcast checks of the distribution constructors.
Mirrors zhusuan/distributions/utils.py:18-71 of the reference (same exception types and messages)."""
import torch
from .._shapes import broadcast_shapes

floating_dtypes = (torch.float32, torch.float16, torch.float64)
log_floating_dtypes = (torch.float32, torch.float64)
integer_dtypes = (torch.int32, torch.int16, torch.int64)


def assert_same_dtype_in(tensors_with_name, dtypes=None):
    """All tensors share one dtype and it is among `dtypes` (zhusuan/distributions/utils.py:12-45)."""
    allowed = set(dtypes) if dtypes else None
    first = None
    for tensor, name in tensors_with_name:
        if allowed and tensor.dtype not in allowed:
            if len(dtypes) == 1:
                raise TypeError('{}({}) must have dtype {}.'.format(name, tensor.dtype, dtypes[0]))
            raise TypeError('{}({}) must have a dtype in {}.'.format(name, tensor.dtype, dtypes))
        if first is None:
            first = (tensor, name)
        elif first[0].dtype != tensor.dtype:
            raise TypeError('{}({}) must have the same dtype as {}({}).'.format(
                name, tensor.dtype, first[1], first[0].dtype))
    return first[0].dtype if first is not None else None


def assert_same_float_dtype(tensors_with_name):
    return assert_same_dtype_in(tensors_with_name, floating_dtypes)


def assert_same_log_float_dtype(tensors_with_name):
    return assert_same_dtype_in(tensors_with_name, log_floating_dtypes)


def check_broadcast(mean, std):
    """RuntimeError when the shapes do not broadcast (the reference evaluates ``mean + std`` for this,
    zhusuan/distributions/utils.py:67-71; only the shapes matter so no kernel is launched here)."""
    broadcast_shapes(tuple(mean.shape), tuple(std.shape))
