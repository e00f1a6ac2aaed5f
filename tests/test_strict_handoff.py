"""The hand-off kernels in two builds that must agree bit for bit.

LJ1 / MS1 (zs_logjoint.hip), PL1 / PM1 / CS1 / AB1 / PR1 (zs_layers.hip), R1 (zs_reinforce.hip) and A1 (zs_adam.hip) end with
"every workgroup writes partials, the last to arrive combines them".  The shipped library orders that hand-off the cheap way
MI355X_MICROARCH.md lists as measured on gfx950 (write-through stores, drained; a barrier; ONE relaxed agent-scope atomic on the
ticket; sc1 loads or one acquire on the consuming side) -- not an architectural guarantee.  `make -C zhusuan-pytorch_amd/csrc strict`
builds the SAME sources with the ticket taken acq_rel at agent scope (buffer_wbl2 / buffer_inv around it: what the HIP memory model
guarantees; tools/_exp/libzs_hip_strict.so, never shipped).  Here whole training steps that are made of those kernels run in both
builds from identical weights, data and Philox streams: every loss of every step and every parameter at the end must be
IDENTICAL.  A stale hand-off in the cheap form is a wrong partial sum: it cannot hide in an equality over thousands of launches.
(K4b, zs_iw.hip, takes its ticket acq_rel in both builds; IW1's batch mean hands nothing over -- each atomic carries data and count.)"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

STRICT = os.path.join(ROOT, "tools", "_exp", "libzs_hip_strict.so")


@pytest.fixture(scope="module")
def builds():
    from zhusuan import _hip
    # the twin is test infrastructure: built HERE when it is missing or older than the sources (not a duty -- and not a
    # failure mode -- of the product's build; `make strict` recompiles only what changed)
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "zhusuan-pytorch_amd", "csrc")
    srcs = glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(ROOT, "include", "zs_hip.h")]
    if not os.path.exists(STRICT) or os.path.getmtime(STRICT) < max(os.path.getmtime(f) for f in srcs):
        r = subprocess.run(["make", "-j6", "-C", csrc, "strict"], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, "make strict failed:\n" + (r.stdout + r.stderr)[-3000:]
    strict = _hip.KernelLibrary(STRICT)
    shipped = _hip.KernelLibrary(_hip.LIB_PATH)
    assert "strict hand-off" in strict.build_info() and "strict" not in shipped.build_info()
    return shipped, strict


class _use(object):
    def __init__(self, klib):
        self.klib = klib

    def __enter__(self):
        from zhusuan import _hip
        self.saved = _hip.lib
        _hip.lib = lambda: self.klib
        return self

    def __exit__(self, *exc):
        from zhusuan import _hip
        _hip.lib = self.saved
        return False


def _train(make, steps, klib):
    """`steps` eagerly launched training steps; returns (losses as float32 bits, final parameters)."""
    import zhusuan as zs
    dev = torch.device("cuda:0")
    with _use(klib):
        torch.manual_seed(11)
        model, obs, opt = make(dev)
        rng = zs.DeviceRNG(dev, seed=77)
        losses = []
        with zs.device_rng(rng):
            for _ in range(steps):
                rng.begin_step()
                for p in model.parameters():
                    p.grad = None
                loss = model(obs)
                loss.backward()
                opt.step()
                losses.append(loss.detach().reshape(1).clone())
        torch.cuda.synchronize()
        out = torch.cat(losses).cpu().numpy().view(np.uint32)
        params = [p.detach().cpu().numpy().copy() for p in model.parameters()]
    return out, params


def _bnn(dev):
    import zhusuan as zs
    from examples import bnn_vi
    model = bnn_vi.build(n_particles=10, device=dev)                      # PM1 both ways, MS1, LJ1, PR1
    g = torch.Generator().manual_seed(5)
    x = torch.randn(512, 13, generator=g).to(dev)
    y = torch.randn(512, generator=g).to(dev)
    return model, {"x": x, "y": y}, zs.optim.FlatAdam(model.parameters(), lr=1e-2)      # A1


def _bnn_per_layer(dev):
    import zhusuan as zs
    from examples import bnn_vi
    model = bnn_vi.build(n_particles=10, device=dev, layer="per_layer")   # PL1 instead of PM1
    g = torch.Generator().manual_seed(6)
    x = torch.randn(300, 13, generator=g).to(dev)
    y = torch.randn(300, generator=g).to(dev)
    return model, {"x": x, "y": y}, zs.optim.FlatAdam(model.parameters(), lr=1e-2)


def _vae(dev):
    import zhusuan as zs
    from examples import vae_mnist
    model = vae_mnist.build(64, hidden=128, device=dev, dense="fused")    # LJ1, MS1 (+ its backward), CS1 / AB1
    x = (torch.rand(64, 784, generator=torch.Generator().manual_seed(7)) < 0.5).float().to(dev)
    return model, {"x": x}, zs.optim.FlatAdam(model.parameters(), lr=1e-3)


def _reinforce(dev):
    import zhusuan as zs
    from zhusuan.framework.bn import BayesianNet
    from zhusuan.variational.elbo import ELBO

    class Q(BayesianNet):
        def __init__(self):
            super().__init__(device=dev)
            self.mu = torch.nn.Parameter(torch.zeros(4096, 8))

        def forward(self, observed):
            self.observe(observed)
            self.normal("z", mean=self.mu, std=torch.ones_like(self.mu.detach()), is_reparameterized=False, reduce_sum_dims=[1])
            return self

    class P(BayesianNet):
        def __init__(self):
            super().__init__(device=dev)        # (a net without parameters: nothing else tells it where its nodes live)

        def forward(self, observed):
            self.observe(observed)
            one = torch.ones(4096, 8, device=dev)
            z = self.normal("z", mean=0 * one, std=one, reduce_sum_dims=[1])
            self.normal("x", mean=z, std=one, reduce_sum_dims=[1])
            return self
    model = ELBO(P(), Q(), estimator="reinforce").to(dev)                  # R1 over 4096 rows: the multi-workgroup form
    x = torch.randn(4096, 8, generator=torch.Generator().manual_seed(8)).to(dev)
    return model, {"x": x}, zs.optim.FlatAdam(model.parameters(), lr=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("scenario,steps", [("bnn", 400), ("bnn_per_layer", 250), ("vae", 250), ("reinforce", 400)])
def test_strict_and_shipped_hand_offs_return_the_same_bits(builds, scenario, steps):
    shipped, strict = builds
    make = {"bnn": _bnn, "bnn_per_layer": _bnn_per_layer, "vae": _vae, "reinforce": _reinforce}[scenario]
    calls = {}
    for name, klib in (("shipped", shipped), ("strict", strict)):
        orig, seen = klib.call, {}

        def spy(entry, *a, _orig=orig, _seen=seen):
            _seen[entry] = _seen.get(entry, 0) + 1
            return _orig(entry, *a)
        klib.call = spy
        try:
            calls[name] = (_train(make, steps, klib), seen)
        finally:
            klib.call = orig
    (l0, p0), seen0 = calls["shipped"]
    (l1, p1), seen1 = calls["strict"]
    assert seen0 == seen1 and sum(seen0.values()) >= 1000, seen0              # the same launches, a thousand and more of them
    want = {"bnn": ("zs_particle_mlp", "zs_normal_sample_logprob_multi", "zs_logjoint_scalar", "zs_adam_step"),
            "bnn_per_layer": ("zs_particle_linear", "zs_logjoint_scalar"),
            "vae": ("zs_logjoint_scalar", "zs_dense_act_bwd", "zs_adam_step"),
            "reinforce": ("zs_reinforce", "zs_adam_step")}[scenario]
    for w in want:
        assert any(k.startswith(w) for k in seen0), (w, sorted(seen0))
    assert np.isfinite(l0.view(np.float32)).all()
    np.testing.assert_array_equal(l0, l1)                                     # every loss of every step, bit for bit
    for a, b in zip(p0, p1):
        np.testing.assert_array_equal(a, b)
    assert len(set(l0.tolist())) > steps // 2                                 # (the steps really differ: fresh draws, moving weights)
