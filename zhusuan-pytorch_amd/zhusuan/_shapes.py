"""Shape helpers on the host path.  torch.broadcast_shapes goes through torch._refs / symbolic-shape
guards and costs ~65 us per call on the host; a step of the IWAE objective needs 13 of them."""


def broadcast_shapes(*shapes):
    """NumPy-style broadcast of plain shapes; RuntimeError when they do not broadcast
    (same exception type as ``mean + std`` in the reference, zhusuan/distributions/utils.py:67-71)."""
    nd = 0
    for s in shapes:
        if len(s) > nd:
            nd = len(s)
    out = [1] * nd
    for s in shapes:
        off = nd - len(s)
        for i, d in enumerate(s):
            d = int(d)
            o = out[off + i]
            if o == 1:
                out[off + i] = d
            elif d != 1 and d != o:
                raise RuntimeError(
                    "Shape mismatch: objects cannot be broadcast to a single shape: %s" % (
                        " vs ".join(str(tuple(int(v) for v in t)) for t in shapes)))
    return tuple(out)


def value_shape(x_shape, lead_ndim, *param_shapes):
    """Shape of ``log_prob`` of a value of shape `x_shape`, with the reference's error behaviour.

    When the value has more axes than the family's first parameter (`lead_ndim` of them), the reference repeats every
    parameter ``[x.shape[0], 1, ..., 1]`` times (normal.py:112-116, bernoulli.py:88-90, logistic.py:73-77,
    uniform.py:73-77) and lets the element-wise arithmetic broadcast.  One extra leading axis is the sample axis; with
    two or more the repeated parameter ``[x.shape[0]] + param.shape`` no longer lines up with the value and the reference
    fails with a RuntimeError from the broadcast (unless the two leading sizes happen to agree).  The kernels here take
    periods instead of copies, so the same check is made on the shapes alone."""
    x_shape = tuple(int(v) for v in x_shape)
    shapes = [x_shape]
    for s in param_shapes:
        s = tuple(int(v) for v in s)
        if len(x_shape) > lead_ndim:
            if len(s) > lead_ndim + 1:
                raise RuntimeError("Number of dimensions of repeat dims can not be smaller than number of dimensions of tensor")
            s = (1,) * (lead_ndim + 1 - len(s)) + s
            s = (x_shape[0] * s[0],) + s[1:]
        shapes.append(s)
    return broadcast_shapes(*shapes)
