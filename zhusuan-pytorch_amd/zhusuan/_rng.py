"""Random-number plumbing for the fused sampling kernels.

Draws use Philox4x32-10 keyed by torch's generator seed for the device; every draw call consumes
one "call id" taken from (and advanced on) that generator's Philox offset, so
``torch.manual_seed(s)`` makes a run reproducible just like it does for torch's own device RNG.

Parity mode: the reference draws its Gaussians on the CPU (zhusuan/distributions/normal.py:102,104),
which no device stream can reproduce, so "identical RNG draws" are obtained by handing the
epsilon tensors in explicitly:

    with zhusuan.inject_epsilon([eps_draw_1, eps_draw_2]):
        loss = model({'x': x})

Each ``Normal._sample`` call pops the next tensor (shape ``[n_samples] + mean.shape``), in the
order the reference would have called ``torch.normal`` -- remember that objectives draw every
latent twice and use the second draw (elbo.py:122, importance_weighted_objective.py:85).
"""
import contextlib

import torch

_queue = None
_host_state = {"seed": 0, "call": 0}


@contextlib.contextmanager
def inject_epsilon(eps_list, strict=True):
    """Supply the standard-normal draws of the enclosed Normal samples explicitly."""
    global _queue
    prev = _queue
    _queue = list(eps_list)
    try:
        yield
        if strict and _queue:
            raise RuntimeError("inject_epsilon: %d epsilon tensors were not consumed" % len(_queue))
    finally:
        _queue = prev


def pop_injected(shape, device):
    """Next injected epsilon (moved to `device`) or None when no injection is active."""
    if _queue is None:
        return None
    if not _queue:
        raise RuntimeError("inject_epsilon: the model drew more Normal samples than epsilons were supplied")
    e = torch.as_tensor(_queue.pop(0), dtype=torch.float32)
    if tuple(e.shape) != tuple(shape):
        raise RuntimeError("inject_epsilon: next epsilon has shape %s, the draw needs %s"
                           % (tuple(e.shape), tuple(shape)))
    return e.to(device).contiguous()


def manual_seed_host(seed):
    """Seed of the host-side (test hook) stream; device streams follow torch.manual_seed."""
    _host_state["seed"] = int(seed)
    _host_state["call"] = 0


def next_call(device):
    """(seed, call id) for one draw on `device`."""
    if device.type == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        gen = torch.cuda.default_generators[idx]
        seed = gen.initial_seed()
        off = gen.get_offset()
        gen.set_offset(off + 4)
        return seed & 0xFFFFFFFFFFFFFFFF, off // 4
    c = _host_state["call"]
    _host_state["call"] = c + 1
    return _host_state["seed"] & 0xFFFFFFFFFFFFFFFF, c
