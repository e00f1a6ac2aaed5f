"""zhusuan.utils: log_mean_exp (zhusuan/utils.py:6-21 of the reference)."""
import torch

from . import _ops

__all__ = ['log_mean_exp']


def log_mean_exp(x, dim=None, keepdims=False):
    """Numerically stable log(mean(exp(x))) over `dim` (int, list/tuple or None = all axes).
    One kernel launch: the reduced axes are moved last and each row is reduced by a wavefront
    (K <= 64) or a workgroup."""
    x = torch.as_tensor(x)
    nd = x.dim()
    if dim is None:
        dims = list(range(nd))
    elif isinstance(dim, (list, tuple)):
        dims = sorted(set(int(d) % nd for d in dim))
    else:
        dims = [int(dim) % nd]
    keep = [d for d in range(nd) if d not in dims]
    perm = keep + dims
    xp = x.permute(perm)
    keep_shape = [x.shape[d] for d in keep]
    K = 1
    for d in dims:
        K *= x.shape[d]
    rows = xp.reshape(-1, K)
    out = _ops.LogMeanExpRows.apply(rows).reshape(keep_shape)
    if keepdims:
        for d in dims:
            out = out.unsqueeze(d)
    return out
