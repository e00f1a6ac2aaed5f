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

Seed-compatible mode: inside ``zhusuan.reference_rng()`` every draw is made on the HOST from
torch's default CPU generator with exactly the call the reference makes at that point, then copied to the device, so

    torch.manual_seed(s)
    with zhusuan.reference_rng():
        loss = model({'x': x})

consumes the CPU stream like the reference run under the same seed (two draws per latent, the second one used) and
yields the reference's numbers.  It costs a host draw and a copy per sample: a reproduction mode, not the fast path.
"""
import contextlib
import contextvars

import torch

# the three switches of this module are context-local (contextvars), like zhusuan.skip_discarded_draws: another thread's
# objective is not affected by a `with` block here
_queue = contextvars.ContextVar("zhusuan_injected_epsilons", default=None)


@contextlib.contextmanager
def inject_epsilon(eps_list, strict=True):
    """Supply the standard-normal draws of the enclosed Normal samples explicitly."""
    q = list(eps_list)
    token = _queue.set(q)
    try:
        yield
        if strict and q:
            raise RuntimeError("inject_epsilon: %d epsilon tensors were not consumed" % len(q))
    finally:
        _queue.reset(token)


_reference_stream = contextvars.ContextVar("zhusuan_reference_rng", default=False)


@contextlib.contextmanager
def reference_rng():
    """Draw like the reference: on the CPU, from torch's default generator (``torch.manual_seed``), call for call."""
    token = _reference_stream.set(True)
    try:
        yield
    finally:
        _reference_stream.reset(token)


def reference_stream_active():
    return _reference_stream.get() and _queue.get() is None


def _host_draw(kind, shape, dtype):
    """The reference's own host call for a draw of `kind`:
    'normal'        torch.normal(0., 1., size=shape)                          normal.py:104  (== mean + std * it for :102)
    'uniform_init'  torch.nn.init.uniform_(torch.empty(shape, dtype), 0, 1)   logistic.py:64
    'rand'          torch.rand(shape, dtype)                                  uniform.py:64,66-67 (torch.distributions.Uniform.sample)"""
    shape = tuple(shape)
    if kind == "normal":
        return torch.normal(0., 1., size=shape).to(dtype)
    if kind == "uniform_init":
        return torch.nn.init.uniform_(torch.empty(shape, dtype=dtype), 0., 1.)
    if kind == "rand":
        return torch.rand(shape, dtype=dtype)
    raise ValueError(kind)


def pop_injected(shape, device, dtype=torch.float32, kind="normal"):
    """Next injected epsilon (moved to `device`, cast to `dtype`); inside ``reference_rng()`` a host draw of `kind`;
    None when neither is active (the kernels then draw from their Philox stream)."""
    q = _queue.get()
    if q is None:
        if _reference_stream.get():
            return _host_draw(kind, shape, dtype).to(device).contiguous()
        return None
    if not q:
        raise RuntimeError("inject_epsilon: the model drew more Normal samples than epsilons were supplied")
    e = torch.as_tensor(q.pop(0), dtype=dtype)
    if tuple(e.shape) != tuple(shape):
        raise RuntimeError("inject_epsilon: next epsilon has shape %s, the draw needs %s"
                           % (tuple(e.shape), tuple(shape)))
    return e.to(device).contiguous()


class DeviceRNG(object):
    """Philox (seed, base offset) kept in DEVICE memory, so that sampling kernels captured in a hipGraph
    draw fresh numbers on every replay: kernels read the state themselves (``rng_state`` argument of the
    C ABI) and every draw inside a step uses ``base + delta`` with a host-side ``delta`` that repeats
    identically on each replay.  ``begin_step()`` (one tiny captured add) moves ``base`` forward.

        rng = zhusuan.DeviceRNG(device, seed=0)
        with zhusuan.device_rng(rng):
            with torch.cuda.graph(g):
                rng.begin_step(); loss = model(obs); loss.backward(); opt.step()
    """

    def __init__(self, device, seed=0, stride=1 << 16):
        self.device = torch.device(device)
        self.state = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=self.device)
        self.stride = int(stride)
        self._delta = 0

    def begin_step(self):
        self.state[1:].add_(self.stride)
        self._delta = 0

    def next_delta(self):
        d = self._delta
        self._delta += 1
        if d >= self.stride:
            raise RuntimeError("DeviceRNG: more than %d draws in one step" % self.stride)
        return d


_device_rng = contextvars.ContextVar("zhusuan_device_rng", default=None)


@contextlib.contextmanager
def device_rng(rng):
    """Route the draws of the enclosed code through `rng` (a DeviceRNG)."""
    token = _device_rng.set(rng)
    try:
        yield rng
    finally:
        _device_rng.reset(token)


# Set by the objectives around their run of the variational net (variational/elbo.py:run_variational): every latent created in
# there is drawn AGAIN by the objective's re-read of node.tensor (elbo.py:122 of the reference), so a Normal node may make
# both draws in one launch and keep the second for that re-read (distributions/normal.py).  Context-local.
_redraw_expected = contextvars.ContextVar("zhusuan_redraw_expected", default=False)
_pair_draws = contextvars.ContextVar("zhusuan_pair_draws", default=True)


@contextlib.contextmanager
def expecting_redraw():
    token = _redraw_expected.set(True)
    try:
        yield
    finally:
        _redraw_expected.reset(token)


@contextlib.contextmanager
def pair_draws(enabled=True):
    """Whether a latent that an objective is about to draw twice makes both draws in ONE launch (default: yes, where the
    sampling kernel takes the shape).  Both draws are executed either way and carry the Philox call ids they would have had
    in two launches when the variational net has one latent; with several latents the ids are handed out per node (first and
    second draw of a node are consecutive) instead of per pass.  ``with zhusuan.pair_draws(False):`` restores one launch per draw."""
    token = _pair_draws.set(bool(enabled))
    try:
        yield
    finally:
        _pair_draws.reset(token)


def pair_draw_wanted():
    return _redraw_expected.get() and _pair_draws.get() and not _reference_stream.get() and _queue.get() is None


def pair_draw_consumable():
    """A pre-made second draw stands in for a fresh one only where a fresh one would have come from the same stream."""
    return not _reference_stream.get() and _queue.get() is None


def next_call(device):
    """(seed, call id, device state tensor or None) for one draw on `device`."""
    rng = _device_rng.get()
    if rng is not None:
        if rng.device != device:
            raise RuntimeError("DeviceRNG lives on %s, draw requested on %s" % (rng.device, device))
        return 0, rng.next_delta(), rng.state
    s, c = _seed_and_call(device)
    return s, c, None


def _seed_and_call(device):
    """(seed, call id) of the next draw from torch's generator of a HIP device: every draw consumes 4 of its offset."""
    if device.type != "cuda":
        raise RuntimeError("zhusuan (MI355X build): draws are made by HIP kernels and device '%s' has none (no CPU path)" % device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    seed = gen.initial_seed()
    off = gen.get_offset()
    gen.set_offset(off + 4)
    return seed & 0xFFFFFFFFFFFFFFFF, off // 4
