"""zhusuan.utils: log_mean_exp (zhusuan/utils.py:6-21 of the reference)."""
import torch

from . import _ops

__all__ = ['log_mean_exp', 'explain', 'warn_on_fallback']


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


# ---------------------------------------------------------------------------------------------------------------------
# Which kernels an objective ran on.  The reference walks its nodes in a Python loop whatever the model
# (zhusuan/variational/importance_weighted_objective.py:66-100, elbo.py:58-79); here an objective takes the one-launch
# kernels when the model has the layout they cover and the per-node kernels otherwise -- same numbers, another speed.
# ``objective.last_path`` says which and why; ``zhusuan.explain(objective)`` prints it; ``zhusuan.warn_on_fallback(True)``
# turns a fallback into a one-time ``warnings.warn`` per reason (off by default).
# ---------------------------------------------------------------------------------------------------------------------
_WARN_ON_FALLBACK = [False]
_WARNED = set()


def warn_on_fallback(on=True):
    """A one-time ``UserWarning`` per (objective class, reason) whenever an objective leaves its one-launch path."""
    _WARN_ON_FALLBACK[0] = bool(on)
    if not on:
        _WARNED.clear()


def note_path(objective, path, why=None):
    """(package-internal) record the path an objective evaluation took."""
    objective.last_path = {"path": path, "why": why}
    if why is not None and _WARN_ON_FALLBACK[0]:
        key = (type(objective).__name__, path, why)
        if key not in _WARNED:
            _WARNED.add(key)
            import warnings
            warnings.warn("zhusuan: %s ran on %s because %s" % (type(objective).__name__, path, why), stacklevel=3)


def explain(objective):
    """The kernels the LAST evaluation of ``objective`` (an ``ELBO`` / ``ImportanceWeightedObjective``) ran on, and -- when
    that was not the one-launch path -- why, as a string.  Evaluate the objective once first."""
    lp = getattr(objective, "last_path", None)
    if lp is None:
        return "%s has not been evaluated yet" % type(objective).__name__
    return lp["path"] if lp["why"] is None else "%s -- because %s" % (lp["path"], lp["why"])
