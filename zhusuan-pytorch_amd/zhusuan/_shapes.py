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
