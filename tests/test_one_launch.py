"""Product-level behaviour of the one-launch paths (LJ1 / MS1 / PL1) and of ``zhusuan.skip_discarded_draws``: what is
launched, and that the fused evaluation equals the node-by-node one (the reference's op sequence) on the same draws.
Runs on the host back-end (package logic over the C oracle) and, marked gpu, on the HIP library."""
import contextlib

import numpy as np
import pytest
import torch

import helpers as H
import zhusuan as zs
from zhusuan import _hip, _ops
from zhusuan.framework.bn import BayesianNet
from zhusuan.framework.stochastic_tensor import LazyDraw
from zhusuan.variational.elbo import ELBO
from zhusuan.variational.importance_weighted_objective import ImportanceWeightedObjective
from examples import bnn_vi, vae_mnist, iwae


@contextlib.contextmanager
def launches():
    """Names of the kernel-library entry points called inside the block, in order."""
    klib = _hip.lib()
    orig, names = klib.call, []

    def spy(name, *a):
        names.append(name[:-4])
        return orig(name, *a)
    klib.call = spy
    try:
        yield names
    finally:
        klib.call = orig


def _bnn(dev, layer, B=16, K=4):
    model = bnn_vi.build(n_particles=K, device=dev, layer=layer)
    wm, wl, yl = H.bnn_params(B, K, device=dev)
    with torch.no_grad():
        for p, v in zip(list(model.variational.w_means) + list(model.variational.w_logstds) + [model.generator.y_logstd],
                        wm + wl + [yl]):
            p.copy_(v)
    x, y, eps = H.bnn_data(B, K)
    return model, {"x": torch.tensor(x, device=dev), "y": torch.tensor(y, device=dev)}, eps


def _grads(model, loss):
    return [g.detach().cpu().numpy() for g in torch.autograd.grad(loss, list(model.parameters()))]


def test_bnn_step_is_one_launch_per_phase_and_equals_the_reference_op_sequence(dev):
    """BNN-VI (examples/bayesian_neural_nets/bnn_vi.py): with the fused network the forward of the objective is
    [the two discarded draws] + ONE sampling launch for both weight matrices + ONE launch for the network + the RMSE diagnostic
    + ONE launch for all five log-probs and the objective; backward mirrors it (three launches).  Value and gradients equal the reference's op sequence (repeat + cat +
    matmul + div + relu, one log-prob kernel per node) on the same epsilons."""
    model, obs, eps = _bnn(dev, "fused")
    with launches() as names, zs.inject_epsilon(eps):
        loss = model(obs)
        n_fwd = len(names)
        g = _grads(model, loss)
    assert names[:n_fwd] == ["zs_normal_sample_logprob"] * 2 + ["zs_normal_sample_logprob_multi", "zs_particle_mlp", "zs_particle_rmse",
                             "zs_logjoint_scalar"]
    assert names[n_fwd:] == ["zs_logjoint_scalar_bwd", "zs_particle_mlp_bwd", "zs_normal_sample_logprob_multi_bwd"]
    # the prior terms read the weight samples through the sampler's alias outputs (indices 4, 5; z: 0, 2; log q: 1, 3),
    # so that their gradient and the network's meet INSIDE the sampler's backward kernel, not in an autograd accumulation launch
    sampler_inputs = sorted(i for f, i in loss.grad_fn.next_functions if f is not None and "NormalSampleLogProbMulti" in type(f).__name__)
    assert sampler_inputs == [4, 5]                   # (the log q rows arrive through a view)
    rmse = float(model.generator.cache["rmse"])
    # one launch per layer (PL1): the same numbers bit for bit
    pl_model, _, _ = _bnn(dev, "per_layer")
    with launches() as pl_names, zs.inject_epsilon(eps):
        pl_loss = pl_model(obs)
        g_pl = _grads(pl_model, pl_loss)
    assert pl_names.count("zs_particle_linear") == 2 and pl_names.count("zs_particle_linear_bwd") == 2
    assert float(pl_loss.detach()) == float(loss.detach()) and all(np.array_equal(a, b) for a, b in zip(g, g_pl))
    ref_model, _, _ = _bnn(dev, "materialize")
    with zs.inject_epsilon(eps):
        ref_loss = ref_model(obs)
    g_ref = _grads(ref_model, ref_loss)
    np.testing.assert_allclose(float(loss.detach()), float(ref_loss.detach()), rtol=3e-6)
    np.testing.assert_allclose(rmse, float(ref_model.generator.cache["rmse"]), rtol=1e-5)
    for a, b in zip(g, g_ref):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-5 * max(np.abs(b).max(), 1e-3))
    # ... and the batched-GEMM formulation of round 2
    bmm_model, _, _ = _bnn(dev, "bmm")
    with zs.inject_epsilon(eps):
        bmm_loss = bmm_model(obs)
    np.testing.assert_allclose(float(loss.detach()), float(bmm_loss.detach()), rtol=3e-6)


def test_skip_discarded_draws_draws_each_latent_once(dev):
    """Inside zhusuan.skip_discarded_draws() the draw that the reference throws away (the node factory's, bn.py:158 /
    elbo.py:122) is not executed: one sampling launch per step instead of three for the BNN, one instead of two for the
    VAE / IWAE; the objective sees exactly the second-draw epsilons, so its value is the default path's."""
    model, obs, eps = _bnn(dev, "fused")
    with zs.inject_epsilon(eps):
        ref = float(model(obs).detach())
    with zs.skip_discarded_draws(), launches() as names, zs.inject_epsilon(eps[2:]):      # only the USED draws are consumed
        loss = model(obs)
    assert names == ["zs_normal_sample_logprob_multi", "zs_particle_mlp", "zs_particle_rmse", "zs_logjoint_scalar"]
    assert float(loss.detach()) == ref
    # VAE: sample + objective = 2 launches forward, objective + sampler backward
    vae = vae_mnist.build(16, hidden=32, device=dev)
    x, e1, e2 = H.vae_data(16)
    xb = torch.tensor(x, device=dev)
    with zs.inject_epsilon([e1, e2]):
        ref = float(vae({"x": xb}).detach())
    with zs.skip_discarded_draws(), launches() as names, zs.inject_epsilon([e2]):
        loss = vae({"x": xb})
        loss.backward()
    assert names == ["zs_normal_sample_logprob", "zs_logjoint_scalar", "zs_logjoint_scalar_bwd", "zs_normal_sample_logprob_bwd"]
    assert float(loss.detach()) == ref
    # IWAE / VIMCO
    m = iwae.build(n_samples=5, estimator="vimco", hidden=32, device=dev)
    x, e1, e2 = H.iwae_data(8, 5)
    xb = torch.tensor(x, device=dev)
    with zs.inject_epsilon([e1, e2]):
        ref = float(m({"x": xb}).detach())
    with zs.skip_discarded_draws(), launches() as names, zs.inject_epsilon([e2]):
        loss = m({"x": xb})
    assert names.count("zs_normal_sample_logprob") == 1 and float(loss.detach()) == ref
    # outside the context nothing changes: two draws again
    with launches() as names, zs.inject_epsilon([e1, e2]):
        m({"x": xb})
    assert names.count("zs_normal_sample_logprob") == 2


def _seed(dev, s):
    if dev.type == "cuda":
        torch.manual_seed(s)
    else:
        import host_backend
        host_backend.manual_seed(s)


def test_both_draws_of_a_latent_in_one_launch_equal_two_launches(dev):
    """Default draws (no injected epsilons): an objective draws every latent twice (stochastic_tensor.py:115-127 + elbo.py:122 of
    the reference).  Where the sampling kernel takes the shape both draws leave in ONE launch (zs_normal_sample_logprob_pair) with
    the Philox call ids two launches use -- value and every gradient are bit for bit those of ``zhusuan.pair_draws(False)``, the
    factory still returns a tensor (the FIRST draw), and the node ends up holding the second."""
    cases = [("iwae-vimco", iwae.build(n_samples=5, estimator="vimco", hidden=32, device=dev), H.iwae_data(8, 5)[0]),
             ("iwae-sgvb", iwae.build(n_samples=5, estimator="sgvb", hidden=32, device=dev), H.iwae_data(8, 5)[0]),
             ("vae", vae_mnist.build(16, hidden=32, device=dev), H.vae_data(16)[0])]
    for label, model, x in cases:
        obs = {"x": torch.tensor(x, device=dev)}

        def run(paired):
            _seed(dev, 11)
            with zs.pair_draws(paired), launches() as names:
                loss = model(obs)
                g = _grads(model, loss)
            return names, float(loss.detach()), g, model.variational.nodes["z"].dist.sample_cache.detach().cpu().numpy()
        n1, l1, g1, z1 = run(True)
        n0, l0, g0, z0 = run(False)
        assert n1.count("zs_normal_sample_logprob_pair") == 1 and n1.count("zs_normal_sample_logprob") == 0, (label, n1)
        assert n0.count("zs_normal_sample_logprob_pair") == 0 and n0.count("zs_normal_sample_logprob") == 2, (label, n0)
        assert l1 == l0 and np.array_equal(z1, z0), label
        for a, b in zip(g1, g0):
            assert np.array_equal(a, b), label
        assert len(n1) == len(n0) - 1, (label, n1, n0)
    # injected epsilons / the reference's host stream / an explicit epsilon: one launch per draw, as before
    m, x = cases[0][1], cases[0][2]
    _, e1, e2 = H.iwae_data(8, 5)
    with launches() as names, zs.inject_epsilon([e1, e2]):
        m({"x": torch.tensor(x, device=dev)})
    assert names.count("zs_normal_sample_logprob") == 2 and names.count("zs_normal_sample_logprob_pair") == 0
    # outside an objective nothing is drawn ahead
    q = m.variational
    with launches() as names:
        q({"x": torch.tensor(x, device=dev)})
    assert names.count("zs_normal_sample_logprob_pair") == 0 and names.count("zs_normal_sample_logprob") == 1
    assert "_pending_draw" not in q.nodes["z"].dist.__dict__


def test_paired_draws_in_a_net_that_reads_its_first_draw(dev):
    """A hierarchical variational net uses the value its first factory returned: with paired draws that value is the first draw of
    the pair (a plain tensor, available at once), the objective scores the second; the second latent's parameters -- computed from
    the first's FIRST draw, as in the reference -- are those of both of ITS draws."""
    class Q(BayesianNet):
        def __init__(self):
            super().__init__()
            self.mu = torch.nn.Parameter(torch.zeros(6, 8))
            self.seen = None

        def forward(self, observed):
            self.observe(observed)
            z1 = self.normal("z1", mean=self.mu, std=torch.ones_like(self.mu.detach()), reduce_mean_dims=[0], reduce_sum_dims=[1])
            self.seen = z1
            self.normal("z2", mean=torch.tanh(z1) * 0.5, std=torch.ones_like(self.mu.detach()), reduce_mean_dims=[0], reduce_sum_dims=[1])
            return self

    class P(BayesianNet):
        def __init__(self):
            super().__init__()
            self.s = torch.nn.Parameter(torch.ones(1))

        def forward(self, observed):
            self.observe(observed)
            one = torch.ones(6, 8, device=self.s.device)
            z1 = self.normal("z1", mean=0 * one, std=one, reduce_mean_dims=[0], reduce_sum_dims=[1])
            z2 = self.normal("z2", mean=z1 * self.s, std=one, reduce_mean_dims=[0], reduce_sum_dims=[1])
            self.normal("x", mean=z2, std=one, reduce_mean_dims=[0], reduce_sum_dims=[1])
            return self
    model = ELBO(P(), Q()).to(dev)
    _seed(dev, 3)
    with launches() as names:
        loss = model({})
        loss.backward()
    q = model.variational
    assert isinstance(q.seen, torch.Tensor) and not isinstance(q.seen, LazyDraw)
    used = q.nodes["z1"].dist.sample_cache
    assert used is not q.seen and not torch.equal(used, q.seen) and torch.isfinite(loss)
    # (two pairs for the latents; the one single draw is the generator's own unobserved node x)
    assert names.count("zs_normal_sample_logprob_pair") == 2 and names.count("zs_normal_sample_logprob") == 1, names
    assert names.count("zs_normal_sample_logprob_multi") == 0             # (the second draws came with the first)
    assert q.mu.grad is not None and torch.isfinite(q.mu.grad).all()
    # the two draws of a pair are independent standard draws: same mean / unit scale statistics
    d = (used - q.mu.detach()).flatten()
    assert abs(float(d.mean())) < 1.5 and 0.2 < float(d.std()) < 3.0


@pytest.mark.parametrize("reparam", [True, False])
def test_in_place_ops_on_a_latent_of_a_paired_draw(dev, reparam):
    """both draws of a pair come out of ONE autograd Function.  Returned as views of shared bases they could
    not be modified in place ("Output 0 of NormalSampleLogProbPairBackward is a view and is being modified inplace ... returns
    multiple views"), which the single-draw path and the reference allow: a variational net that does z.mul_() / z += ... on
    the value its factory returned must work with pair_draws on (the default) exactly as with pair_draws(False)."""
    class Q(BayesianNet):
        def __init__(self):
            super().__init__(device=dev)
            self.mu = torch.nn.Parameter(torch.full((6, 8), 0.25))
            self.ls = torch.nn.Parameter(torch.full((6, 8), -0.5))
            self.two = False

        def forward(self, observed):
            self.observe(observed)
            z1 = self.normal("z1", mean=self.mu, logstd=self.ls, is_reparameterized=reparam, reduce_mean_dims=[0], reduce_sum_dims=[1])
            z1.mul_(0.5)                  # in place, on the first draw of the pair
            z1 += 1.0
            z1.clamp_(-3.0, 3.0)
            if self.two:
                self.normal("z2", mean=z1 * self.mu, std=torch.ones_like(self.mu.detach()), is_reparameterized=reparam,
                            reduce_mean_dims=[0], reduce_sum_dims=[1])
            return self

    class P(BayesianNet):
        def __init__(self):
            super().__init__(device=dev)      # (a net without parameters: nothing else tells it where its nodes live)
            self.two = False

        def forward(self, observed):
            self.observe(observed)
            one = torch.ones(6, 8, device=dev)
            z1 = self.normal("z1", mean=0 * one, std=one, reduce_mean_dims=[0], reduce_sum_dims=[1])
            z2 = self.normal("z2", mean=z1, std=one, reduce_mean_dims=[0], reduce_sum_dims=[1]) if self.two else z1
            self.normal("x", mean=z2, std=one, reduce_mean_dims=[0], reduce_sum_dims=[1])
            return self
    # a hierarchical net whose second latent is computed from the modified first draw: runs, finite gradients
    model = ELBO(P(), Q(), estimator="sgvb").to(dev)
    model.generator.two = model.variational.two = True
    _seed(dev, 11)
    with zs.pair_draws(True), launches() as names:
        loss = model({"x": torch.zeros(6, 8, device=dev)})
        loss.backward()
    assert names.count("zs_normal_sample_logprob_pair") == 2 and torch.isfinite(loss) and torch.isfinite(model.variational.mu.grad).all()
    # one latent: the pair's draws carry the call ids of two launches, so value and gradients equal pair_draws(False)
    results = {}
    for paired in (True, False):
        model = ELBO(P(), Q(), estimator="sgvb").to(dev)
        model.generator.two = model.variational.two = False
        _seed(dev, 11)
        with zs.pair_draws(paired), launches() as names:
            loss = model({"x": torch.zeros(6, 8, device=dev)})
            loss.backward()
        assert (names.count("zs_normal_sample_logprob_pair") > 0) == paired, names
        q = model.variational
        results[paired] = (float(loss.detach()), q.mu.grad.detach().cpu().clone(), q.ls.grad.detach().cpu().clone())
        assert torch.isfinite(loss) and torch.isfinite(q.mu.grad).all()
    np.testing.assert_allclose(results[True][0], results[False][0], rtol=1e-6)
    np.testing.assert_allclose(results[True][1].numpy(), results[False][1].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(results[True][2].numpy(), results[False][2].numpy(), rtol=1e-5, atol=1e-7)
    # the second draw of the pair (the one the objective scores) may be modified in place as well -- and, for a draw that is not
    # reparameterised, autograd must NOTICE it (the K-summed backward needs the value): same contract as the single draw
    mu = torch.full((6, 8), 0.25, device=dev, requires_grad=True)
    sg = torch.ones(6, 8, device=dev)
    with zs.pair_draws(True):
        from zhusuan import _rng
        with _rng.expecting_redraw():
            d = zs.distributions.Normal(mean=mu, std=sg, is_reparameterized=reparam, group_ndims=1)     # (fused log-density)
            first = d.sample(1)
            assert "_pending_draw" in d.__dict__
            second = d.sample(1)
    assert first.data_ptr() != second.data_ptr() and first._base is None and second._base is None
    lp = d.log_prob(second)
    assert lp is d._fused[1] and lp._base is None
    second.mul_(2.0)
    if reparam:
        (lp.sum() + second.sum()).backward()
        assert torch.isfinite(mu.grad).all()
    else:
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            lp.sum().backward()


class _Hierarchical(BayesianNet):
    """q(z1) q(z2 | z1): the net's own code READS the value its first node factory returned."""

    def __init__(self):
        super().__init__()
        self.mu = torch.nn.Parameter(torch.zeros(6, 3))
        self.seen = None

    def forward(self, observed):
        self.observe(observed)
        z1 = self.normal("z1", mean=self.mu, std=torch.ones_like(self.mu.detach()), reduce_mean_dims=[0], reduce_sum_dims=[1])
        self.seen = z1
        m2 = torch.tanh(z1) * 0.5 + z1[:, :1] - (1.0 - z1).mean()           # torch function, indexing, reflected arithmetic, method
        self.normal("z2", mean=m2, std=torch.ones_like(self.mu.detach()), reduce_mean_dims=[0], reduce_sum_dims=[1])
        return self


class _Gen(BayesianNet):
    def __init__(self):
        super().__init__()
        self.s = torch.nn.Parameter(torch.ones(1))

    def forward(self, observed):
        self.observe(observed)
        z1 = self.normal("z1", mean=torch.zeros(6, 3, device=self.s.device), std=torch.ones(6, 3, device=self.s.device),
                         reduce_mean_dims=[0], reduce_sum_dims=[1])
        z2 = self.normal("z2", mean=z1 * self.s, std=torch.ones(6, 3, device=self.s.device), reduce_mean_dims=[0], reduce_sum_dims=[1])
        self.normal("x", mean=z2, std=torch.ones(6, 3, device=self.s.device), reduce_mean_dims=[0], reduce_sum_dims=[1])
        return self


def test_a_net_that_reads_its_first_draw_still_gets_it(dev):
    """Default (switch off): the factory returns a tensor, as in the reference.  Switch on: it returns a LazyDraw that
    samples the moment the net's code touches it -- the value the reference's factory would have returned -- so a
    hierarchical variational net computes the same numbers either way."""
    rng = np.random.RandomState(3)
    eps = [rng.standard_normal((6, 3)).astype(np.float32) for _ in range(4)]       # z1#1, z2#1, z1#2, z2#2
    x = torch.tensor(rng.standard_normal((6, 3)).astype(np.float32), device=dev)
    model = ELBO(_Gen(), _Hierarchical()).to(dev)
    with zs.inject_epsilon(eps):
        ref = model({"x": x})
    assert isinstance(model.variational.seen, torch.Tensor)
    g_ref = _grads(model, ref)
    with zs.skip_discarded_draws(), launches() as names, zs.inject_epsilon(eps[:3]):
        loss = model({"x": x})
    assert isinstance(model.variational.seen, LazyDraw) and "drawn" in repr(model.variational.seen)
    # z1 was touched (drawn inside the net: eps[0]), z2 was not (its discarded draw is skipped), then the objective re-read
    # z1 (eps[1]) and z2 (eps[2]): three draws.  The default path makes the same three plus z2's discarded one (any epsilon)
    assert names.count("zs_normal_sample_logprob") == 1 and names.count("zs_normal_sample_logprob_multi") == 1   # (re-read: one launch)
    with zs.inject_epsilon([eps[0], eps[3], eps[1], eps[2]]):
        same = model({"x": x})
    np.testing.assert_allclose(float(loss.detach()), float(same.detach()), rtol=1e-6)
    g = _grads(model, loss)
    assert all(np.isfinite(a).all() for a in g) and len(g) == len(g_ref)


def test_multi_sampler_draws_what_the_node_by_node_loop_draws(dev):
    """MS1 consumes the same Philox call ids as the per-node loop: the BNN's two weight matrices come out identical
    whether they are drawn by one launch or by two (same seed), and so does the objective."""
    import host_backend
    from zhusuan.variational import elbo as elbo_mod
    model, obs, _ = _bnn(dev, "fused")

    def run(batched):
        if dev.type == "cuda":
            torch.manual_seed(5)
        else:
            host_backend.manual_seed(5)
        limit = elbo_mod._MULTI_DRAW_MAX_ELEMENTS
        elbo_mod._MULTI_DRAW_MAX_ELEMENTS = limit if batched else 0
        try:
            with launches() as names:
                loss = model(obs)
        finally:
            elbo_mod._MULTI_DRAW_MAX_ELEMENTS = limit
        return names, float(loss.detach()), [model.variational.nodes[n].dist.sample_cache.detach().cpu().numpy() for n in ("w0", "w1")]
    n1, l1, w1 = run(True)
    n2, l2, w2 = run(False)
    assert n1.count("zs_normal_sample_logprob_multi") == 1 and n2.count("zs_normal_sample_logprob_multi") == 0
    assert n2.count("zs_normal_sample_logprob") == 4
    assert all(np.array_equal(a, b) for a, b in zip(w1, w2))
    np.testing.assert_allclose(l1, l2, rtol=2e-6)


def test_objective_with_more_nodes_than_the_table_falls_back(dev):
    """Nine scalar nodes do not fit LJ1's table of eight: the objective takes the node-by-node path (same value)."""
    class Many(BayesianNet):
        def __init__(self, n):
            super().__init__()
            self.n = n
            self.p = torch.nn.Parameter(torch.zeros(4))

        def forward(self, observed):
            self.observe(observed)
            for i in range(self.n):
                self.normal("z%d" % i, mean=self.p + i, std=torch.ones(4, device=self.p.device), reduce_sum_dims=[0])
            return self
    rng = np.random.RandomState(0)
    eps = [rng.standard_normal(4).astype(np.float32) for _ in range(10)]
    for n, fused in ((3, True), (5, False)):
        m = ELBO(Many(n), Many(n)).to(dev)
        with launches() as names, zs.inject_epsilon(eps[:2 * n]):
            loss = m({})
        assert ("zs_logjoint_scalar" in names) == fused
        # -(sum_p log p(z) - sum_q log q(z)) with identical p and q at the same z: exactly zero either way
        assert abs(float(loss.detach())) < 1e-5


def test_large_objectives_keep_the_streaming_kernels(dev):
    """Beyond the launch-bound regime the nodes' own log-prob kernels run (K2 / K3) and LJ1 only adds their rows up."""
    from zhusuan.variational import elbo as elbo_mod
    vae = vae_mnist.build(16, hidden=32, device=dev)
    x, e1, e2 = H.vae_data(16)
    xb = torch.tensor(x, device=dev)
    with zs.inject_epsilon([e1, e2]):
        ref = vae({"x": xb})
    g_ref = _grads(vae, ref)
    limit = elbo_mod._LOGJOINT_MAX_ELEMENTS
    elbo_mod._LOGJOINT_MAX_ELEMENTS = 1000          # the Bernoulli term alone has 16 * 784 elements
    try:
        with launches() as names, zs.inject_epsilon([e1, e2]):
            loss = vae({"x": xb})
            g = _grads(vae, loss)
    finally:
        elbo_mod._LOGJOINT_MAX_ELEMENTS = limit
    assert "zs_bernoulli_logprob" in names and "zs_normal_logprob" in names and names.count("zs_logjoint_scalar") == 1
    np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=2e-6)
    for a, b in zip(g, g_ref):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6 * max(np.abs(b).max(), 1))


def test_particle_linear_layer_against_torch(dev):
    rng = np.random.RandomState(8)
    for (K, B, n_in, n_out, shared, relu) in [(4, 16, 13, 50, True, True), (4, 16, 50, 1, False, False), (3, 70, 300, 5, False, True)]:
        h = torch.tensor(rng.standard_normal((B, n_in) if shared else (K, B, n_in)).astype(np.float32), device=dev, requires_grad=True)
        w = torch.tensor(rng.standard_normal((K, n_out, n_in + 1)).astype(np.float32), device=dev, requires_grad=True)
        with launches() as names:
            out = zs.particle_linear(h, w, relu=relu)
            gh, gw = torch.autograd.grad((out * torch.linspace(-1, 1, out.numel(), device=dev).view_as(out)).sum(), [h, w])
        assert (names == ["zs_particle_linear", "zs_particle_linear_bwd"]) == (n_in <= 255)     # larger layers: torch's batched GEMM
        hd, wd = h.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
        hh = hd.unsqueeze(0).expand(K, B, n_in) if shared else hd
        ref = (torch.bmm(hh, wd[:, :, :n_in].transpose(1, 2)) + wd[:, :, n_in].unsqueeze(1)) / np.sqrt(n_in + 1)
        ref = torch.relu(ref) if relu else ref
        rh, rw = torch.autograd.grad((ref * torch.linspace(-1, 1, ref.numel(), dtype=torch.float64).view_as(ref)).sum(), [hd, wd])
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(gh.cpu().numpy(), rh.numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(gw.cpu().numpy(), rw.numpy(), rtol=2e-4, atol=2e-4 * max(float(rw.abs().max()), 1))
    with pytest.raises(RuntimeError, match="does not match"):
        zs.particle_linear(torch.zeros(2, 3, 4, device=dev), torch.zeros(2, 5, 7, device=dev))


def test_zhusuan_linear_is_nn_linear_with_a_one_launch_bias_gradient(dev):
    """zhusuan.Linear: same parameters, same forward, same gradients as torch.nn.Linear; the bias gradient comes from CS1."""
    torch.manual_seed(0)
    ref = torch.nn.Linear(37, 23).to(dev)
    lin = zs.Linear(37, 23).to(dev)
    lin.load_state_dict(ref.state_dict())
    assert [n for n, _ in lin.named_parameters()] == [n for n, _ in ref.named_parameters()]
    for shape in [(50, 37), (4, 11, 37)]:
        x1 = torch.randn(*shape, device=dev, requires_grad=True)
        x2 = x1.detach().clone().requires_grad_(True)
        with launches() as names:
            y = lin(x1)
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
        assert names == ["zs_column_sum"]
        yr = ref(x2)
        (yr * torch.linspace(-1, 1, yr.numel(), device=dev).view_as(yr)).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(lin.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(lin.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
        lin.zero_grad(); ref.zero_grad()
    with torch.no_grad():
        assert torch.equal(lin(x1), ref(x1))
    # the example callers accept it: same objective value as with torch.nn.Linear on the same weights and draws
    x, e1, e2 = H.vae_data(16)
    xb = torch.tensor(x, device=dev)
    ma, mb = vae_mnist.build(16, hidden=32, device=dev, dense='torch'), vae_mnist.build(16, hidden=32, device=dev, dense='zhusuan')
    mb.load_state_dict(ma.state_dict())
    with zs.inject_epsilon([e1, e2]):
        la = ma({"x": xb})
    with zs.inject_epsilon([e1, e2]):
        lb = mb({"x": xb})
    ga, gb = _grads(ma, la), _grads(mb, lb)
    assert float(la.detach()) == float(lb.detach())
    for a, b in zip(ga, gb):
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-6 * max(np.abs(a).max(), 1))


def test_zhusuan_sequential_fuses_linear_with_its_activation(dev):
    """zhusuan.Sequential: torch.nn.Sequential's children, indices, names and slices; every zhusuan.Linear -> ReLU / Sigmoid pair
    runs as one fused layer (ReLU in the GEMM's epilogue; activation backward + bias gradient = AB1).  Same value and gradients
    as torch.nn's modules on the same weights."""
    torch.manual_seed(1)
    ref = torch.nn.Sequential(torch.nn.Linear(19, 24), torch.nn.ReLU(), torch.nn.Linear(24, 12), torch.nn.ReLU(),
                              torch.nn.Linear(12, 8), torch.nn.Sigmoid()).to(dev)
    seq = zs.Sequential(zs.Linear(19, 24), torch.nn.ReLU(), zs.Linear(24, 12), torch.nn.ReLU(), zs.Linear(12, 8),
                        torch.nn.Sigmoid()).to(dev)
    seq.load_state_dict(ref.state_dict())
    assert [n for n, _ in seq.named_parameters()] == [n for n, _ in ref.named_parameters()]
    assert type(seq[:-1]) is zs.Sequential and len(seq[:-1]) == 5
    for shape in [(33, 19), (3, 7, 19)]:
        x1 = torch.randn(*shape, device=dev, requires_grad=True)
        x2 = x1.detach().clone().requires_grad_(True)
        with launches() as names:
            y = seq(x1)
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
        assert names == ["zs_dense_act_bwd"] * 3
        yr = ref(x2)
        (yr * torch.linspace(-1, 1, yr.numel(), device=dev).view_as(yr)).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
        for (n, a), (_, b) in zip(seq.named_parameters(), ref.named_parameters()):
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-5, atol=1e-5, err_msg=n)
        seq.zero_grad(); ref.zero_grad()
        with launches() as names:                                              # the slice (logits: no Sigmoid) keeps fusing
            yl = seq[:-1](x1)
            yl.sum().backward()
        assert names == ["zs_dense_act_bwd"] * 2 + ["zs_column_sum"] or names == ["zs_column_sum"] + ["zs_dense_act_bwd"] * 2
        np.testing.assert_allclose(yl.detach().cpu().numpy(), ref[:-1](x2).detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
        seq.zero_grad()
    with torch.no_grad():
        np.testing.assert_allclose(seq(x1).cpu().numpy(), ref(x1).cpu().numpy(), rtol=1e-6, atol=1e-7)
    # own activation argument; without bias; a layer that is not followed by an activation; a foreign module in between
    lin = zs.Linear(19, 5, bias=False, activation='relu').to(dev)
    xa = torch.randn(9, 19, device=dev, requires_grad=True)
    out = lin(xa)
    out.sum().backward()
    with torch.no_grad():
        assert torch.equal(out, torch.relu(xa @ lin.weight.t()))
    wref = lin.weight.detach().clone().requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    torch.relu(xb @ wref.t()).sum().backward()
    np.testing.assert_allclose(lin.weight.grad.cpu().numpy(), wref.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError, match="activation"):
        zs.Linear(3, 3, activation='tanh')
    # precisions without kernels take torch's ops (same numbers as torch.nn's modules), with autograd
    if dev.type == "cuda":
        half = zs.Sequential(zs.Linear(19, 6), torch.nn.ReLU()).to(dev).to(torch.bfloat16)
        xh = torch.randn(9, 19, device=dev, dtype=torch.bfloat16, requires_grad=True)
        with launches() as names:
            half(xh).sum().backward()
        assert names == [] and xh.grad is not None and half[0].bias.grad is not None
    mixed = zs.Sequential(zs.Linear(19, 6), torch.nn.Tanh(), torch.nn.Linear(6, 4), torch.nn.ReLU()).to(dev)
    with launches() as names:
        mixed(xa).sum().backward()
    assert names == ["zs_column_sum"]
    # the example callers: dense='fused' equals dense='torch' on the same weights and draws
    x, e1, e2 = H.vae_data(16)
    xt = torch.tensor(x, device=dev)
    ma, mb = vae_mnist.build(16, hidden=32, device=dev, dense='torch'), vae_mnist.build(16, hidden=32, device=dev, dense='fused')
    mb.load_state_dict(ma.state_dict())
    with zs.inject_epsilon([e1, e2]):
        la = ma({"x": xt})
    with zs.inject_epsilon([e1, e2]):
        lb = mb({"x": xt})
    ga, gb = _grads(ma, la), _grads(mb, lb)
    np.testing.assert_allclose(float(lb.detach()), float(la.detach()), rtol=1e-6)
    for a, b in zip(ga, gb):
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-6 * max(np.abs(a).max(), 1))


def test_particle_mlp_is_the_layer_chain_in_one_launch(dev):
    """zhusuan.particle_mlp: up to four layers that fit the LDS = one launch each way (PM1), bit-identical to the chain of
    zhusuan.particle_linear calls; deeper / wider networks take the chain."""
    for sizes, K, B, shared, fused in [((13, 50, 1), 4, 33, True, True), ((6, 10, 10, 3), 3, 20, False, True),
                                       ((4, 4, 4, 4, 4, 2), 2, 9, True, False), ((300, 8, 2), 2, 5, True, False)]:
        torch.manual_seed(len(sizes) + B)
        x = torch.randn(*((B, sizes[0]) if shared else (K, B, sizes[0])), device=dev, requires_grad=True)
        ws = [torch.randn(K, sizes[l + 1], sizes[l] + 1, device=dev, requires_grad=True) for l in range(len(sizes) - 1)]
        with launches() as names:
            y = zs.particle_mlp(x, ws)
            coef = torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)
            g = torch.autograd.grad((y * coef).sum(), [x] + ws)
        assert (names == ["zs_particle_mlp", "zs_particle_mlp_bwd"]) == fused, names
        h = x
        for l, w in enumerate(ws):
            h = zs.particle_linear(h, w, relu=l < len(ws) - 1)
        g_ref = torch.autograd.grad((h * coef).sum(), [x] + ws)
        assert torch.equal(y, h)
        for a, b in zip(g[1:], g_ref[1:]):
            assert torch.equal(a, b)
        np.testing.assert_allclose(g[0].cpu().numpy(), g_ref[0].cpu().numpy(), rtol=1e-5, atol=1e-6)     # (shared x: summed over K by torch)
    with pytest.raises(RuntimeError, match="does not match"):
        zs.particle_mlp(torch.zeros(3, 4, device=dev), [torch.zeros(2, 5, 5, device=dev), torch.zeros(2, 2, 7, device=dev)])
    with pytest.raises(ValueError, match="at least one layer"):
        zs.particle_mlp(torch.zeros(3, 4, device=dev), [])


def test_layer_functions_accept_deferred_node_values(dev):
    """Inside zhusuan.skip_discarded_draws() a variational net's node values are LazyDraw handles; the layer functions of this
    package draw them like any torch function would."""
    class Q(BayesianNet):
        def __init__(self):
            super().__init__()
            self.mu = torch.nn.Parameter(torch.zeros(3, 5, 4))
            self.fc = zs.Linear(4, 2)

        def forward(self, observed):
            self.observe(observed)
            w = self.normal(name="w", mean=self.mu, std=torch.ones_like(self.mu), group_ndims=2, reduce_mean_dims=None)
            self.cache["is_lazy"] = isinstance(w, LazyDraw)
            self.cache["a"] = zs.particle_linear(torch.ones(6, 3, device=self.mu.device), w, relu=True)
            self.cache["b"] = zs.particle_mlp(torch.ones(6, 3, device=self.mu.device), [w, w[:, :2, :].contiguous().repeat(1, 1, 2)[:, :, :6]])
            self.cache["c"] = self.fc(w)
            self.cache["d"] = zs.particle_rmse(self.cache["a"][:, :, 0], torch.zeros(6, device=self.mu.device))
            return self
    q = Q().to(dev)
    with zs.skip_discarded_draws(), zs.framework.stochastic_tensor.deferred_node_values():
        q({})
    assert q.cache["is_lazy"] and q.cache["a"].shape == (3, 6, 5) and q.cache["b"].shape == (3, 6, 2)
    assert q.cache["c"].shape == (3, 5, 2) and q.cache["d"].dim() == 0


def test_skip_discarded_draws_is_context_local_and_stale_handles_raise(dev):
    """The switch is a contextvar: a thread that evaluates its own objective while another one sits inside
    ``zhusuan.skip_discarded_draws()`` keeps the default.  A LazyDraw that user code stored and touches
    only AFTER the objective has re-read the node stands for the draw the reference discards: it raises instead of drawing
    out of order."""
    import threading
    from zhusuan.framework import stochastic_tensor as st
    seen = {}

    def other():
        seen["skip"] = st.skipping_discarded_draws()
        with zs.skip_discarded_draws():
            seen["inner"] = st.skipping_discarded_draws()
    with zs.skip_discarded_draws():
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert st.skipping_discarded_draws() is True
    assert seen == {"skip": False, "inner": True} and st.skipping_discarded_draws() is False

    class Q(BayesianNet):
        def __init__(self):
            super().__init__()
            self.mu = torch.nn.Parameter(torch.zeros(4, 3))

        def forward(self, observed):
            self.observe(observed)
            self.kept = self.normal(name="z", mean=self.mu, std=torch.ones_like(self.mu), reduce_mean_dims=[0], reduce_sum_dims=[1])
            return self

    class P(BayesianNet):
        def forward(self, observed):
            self.observe(observed)
            dev_ = self.observed["z"].device
            self.normal(name="z", mean=torch.zeros(4, 3, device=dev_), std=torch.ones(4, 3, device=dev_), reduce_mean_dims=[0],
                        reduce_sum_dims=[1])
            return self
    from zhusuan.variational.elbo import ELBO
    q = Q().to(dev)
    model = ELBO(P().to(dev), q)
    with zs.skip_discarded_draws(), launches() as names:
        loss = model({})
        assert names.count("zs_normal_sample_logprob") == 1 and isinstance(q.kept, LazyDraw)
        with pytest.raises(RuntimeError, match="was not used before the objective drew the node again"):
            q.kept + 1.0
        assert "expired" in repr(q.kept)
    assert torch.isfinite(loss)
    loss = model({})                    # outside the context the factory returns the (discarded) first draw, as in the reference
    assert isinstance(q.kept, torch.Tensor) and q.kept.shape == (4, 3)
