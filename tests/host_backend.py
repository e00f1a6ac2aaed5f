"""TEST INFRASTRUCTURE: run the zhusuan package's HOST LOGIC (shapes, reductions, autograd wiring, ctypes marshalling)
on a GPU-less machine by pointing its kernel calls at the plain-C oracle library (oracle/zs_oracle_c.c: the same C ABI
on host pointers).

The package itself contains no such routing (zhusuan/_hip.py raises for any tensor that is not on a HIP device, and
for a missing libzs_hip.so).  Everything that makes CPU tensors acceptable lives HERE, as monkeypatches applied by
tests only:

    _hip.lib             -> the oracle library
    _hip.require_device  -> accepts CPU tensors only (a GPU tensor under the host back-end is a test bug)
    _hip.default_device  -> cpu
    _rng._seed_and_call  -> a host-side (seed, call counter) pair instead of torch's device generator

``install(klib)`` / ``uninstall()`` are idempotent; ``active()`` tells whether the patches are in place."""
import torch

_saved = None
_state = {"seed": 0, "call": 0}


def active():
    return _saved is not None


def manual_seed(seed):
    """Seed of the host-side Philox (seed, call) pair used while the host back-end is installed."""
    _state["seed"] = int(seed)
    _state["call"] = 0


def install(klib):
    global _saved
    from zhusuan import _hip, _rng
    if _saved is None:
        _saved = (_hip.lib, _hip.require_device, _hip.default_device, _rng._seed_and_call)

    def require_device(*tensors):
        for t in tensors:
            if t is not None and t.device.type != "cpu":
                raise RuntimeError("tests/host_backend: host library installed but tensor is on %s" % t.device)
        return None

    def seed_and_call(device):
        if device.type != "cpu":
            raise RuntimeError("tests/host_backend: draw requested on %s" % device)
        c = _state["call"]
        _state["call"] = c + 1
        return _state["seed"] & 0xFFFFFFFFFFFFFFFF, c

    _hip.lib = lambda: klib
    _hip.require_device = require_device
    _hip.default_device = lambda: torch.device("cpu")
    _rng._seed_and_call = seed_and_call


def uninstall():
    global _saved
    if _saved is None:
        return
    from zhusuan import _hip, _rng
    _hip.lib, _hip.require_device, _hip.default_device, _rng._seed_and_call = _saved
    _saved = None
