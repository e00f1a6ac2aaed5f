"""zhusuan.GraphedStep: a captured training step must train exactly like the eager one (same device-resident Philox
state, same optimizer), with and without an eager exchange between two graphs; constructing it with restore=True has
no side effect on parameters, optimizer state or RNG state."""
import numpy as np
import pytest
import torch

import zhusuan as zs
from examples import vae_mnist, iwae


def test_graphed_step_needs_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        zs.GraphedStep(lambda: None)


def _make(kind, dev, seed=5):
    torch.manual_seed(seed)
    if kind == "vae":
        model = vae_mnist.build(32, device=dev)
        B = 32
    else:
        model = iwae.build(5, "vimco", hidden=64, device=dev)
        B = 16
    x = (torch.rand(B, 784, device=dev) < 0.5).float()
    opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True, capturable=True)
    rng = zs.DeviceRNG(dev, seed=123)
    return model, opt, rng, {"x": x}


def _compute(model, rng, obs):
    def compute():
        rng.begin_step()
        for p in model.parameters():
            p.grad = None
        loss = model(obs)
        loss.backward()
        return loss.detach()
    return compute


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["vae", "iwae"])
@pytest.mark.parametrize("with_exchange", [False, True])
def test_graphed_step_matches_eager(kind, with_exchange):
    dev = torch.device("cuda:0")
    model_e, opt_e, rng_e, obs = _make(kind, dev)
    model_g, opt_g, rng_g, _ = _make(kind, dev)
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        assert torch.equal(pe, pg)
    calls = []

    def exchange(loss):
        calls.append(1)
        return loss
    before = [p.detach().clone() for p in model_g.parameters()]
    step = zs.GraphedStep(_compute(model_g, rng_g, obs), opt_g.step, exchange=exchange if with_exchange else None, rng=rng_g,
                          warmup=3, restore=True)
    assert len(step.graphs) == (2 if with_exchange else 1)
    for b, p in zip(before, model_g.parameters()):
        assert torch.equal(b, p), "restore=True must undo the warm-up steps"
    assert torch.equal(rng_g.state, rng_e.state)
    calls.clear()
    comp_e = _compute(model_e, rng_e, obs)
    le, lg = [], []
    with zs.device_rng(rng_e):
        for _ in range(6):
            le.append(float(comp_e()))
            opt_e.step()
    for _ in range(6):
        lg.append(float(step()))
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    assert len(set(lg)) == 6                                   # fresh draws on every replay
    if with_exchange:
        assert len(calls) == 6
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["vae", "iwae"])
def test_several_steps_per_replay_equal_the_same_steps_one_by_one(kind):
    """GraphedStep(steps_per_replay=4): one graph launch = four training steps, each with fresh draws and its own Adam step;
    three replays equal twelve eager steps."""
    dev = torch.device("cuda:0")
    model_e, opt_e, rng_e, obs = _make(kind, dev)
    model_g, opt_g, rng_g, _ = _make(kind, dev)
    step = zs.GraphedStep(_compute(model_g, rng_g, obs), opt_g.step, rng=rng_g, warmup=3, restore=True, steps_per_replay=4)
    assert len(step.graphs) == 1 and torch.equal(rng_g.state, rng_e.state)
    comp_e = _compute(model_e, rng_e, obs)
    le, lg = [], []
    with zs.device_rng(rng_e):
        for _ in range(12):
            le.append(float(comp_e()))
            opt_e.step()
    for _ in range(3):
        lg.append(float(step()))
    np.testing.assert_allclose(lg, le[3::4], rtol=2e-5)        # the loss a replay returns is its last sub-step's
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=2e-4, atol=2e-6)
    with pytest.raises(ValueError, match="steps_per_replay"):
        zs.GraphedStep(_compute(model_g, rng_g, obs), opt_g.step, exchange=lambda l: l, rng=rng_g, steps_per_replay=2)


@pytest.mark.gpu
def test_graphed_forward_only():
    dev = torch.device("cuda:0")
    model, _, rng, obs = _make("iwae", dev)

    def compute():
        rng.begin_step()
        with torch.no_grad():
            return model(obs).detach()
    step = zs.GraphedStep(compute, None, rng=rng)
    vals = [float(step()) for _ in range(4)]
    assert len(set(vals)) == 4 and all(np.isfinite(vals))


def _make_bnn(dev, seed=5):
    from examples import bnn_vi
    torch.manual_seed(seed)
    model = bnn_vi.build(n_particles=6, device=dev)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.1 * torch.randn_like(p))
    obs = {"x": torch.randn(96, 13, device=dev), "y": torch.randn(96, device=dev)}
    opt = zs.optim.FlatAdam(model.parameters(), lr=1e-2)
    return model, opt, zs.DeviceRNG(dev, seed=77), obs


@pytest.mark.gpu
@pytest.mark.parametrize("skip", [False, True])
def test_one_launch_paths_replay_from_a_graph(skip):
    """The BNN step -- MS1 (both weight matrices in one sampling launch), PL1 (its two layers), LJ1 (all five log-probs and the
    objective), A1 -- captured in one hipGraph trains exactly like the eager step, with and without the discarded draws; the
    workspaces and tickets of the hand-offs inside LJ1 / PL1 survive capture and replay."""
    dev = torch.device("cuda:0")
    model_e, opt_e, rng_e, obs = _make_bnn(dev)
    model_g, opt_g, rng_g, _ = _make_bnn(dev)
    with zs.skip_discarded_draws(skip):
        step = zs.GraphedStep(_compute(model_g, rng_g, obs), opt_g.step, rng=rng_g, warmup=3, restore=True)
        comp_e = _compute(model_e, rng_e, obs)
        le, lg = [], []
        with zs.device_rng(rng_e):
            for _ in range(5):
                le.append(float(comp_e()))
                opt_e.step()
        for _ in range(5):
            lg.append(float(step()))
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    assert len(set(lg)) == 5
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_staged_graphs_with_the_scalar_objective():
    """zhusuan.GraphedStages (the multi-rank step's shape: backward in two stages around eagerly launched collectives) over
    the VAE, whose objective is ONE LJ1 launch: stage 2 runs LJ1's backward a second time on the retained graph.  Equal to
    the eager staged step."""
    from zhusuan import dataparallel
    dev = torch.device("cuda:0")

    def make():
        torch.manual_seed(3)
        model = vae_mnist.build(32, hidden=64, device=dev, dense="zhusuan")
        x = (torch.rand(32, 784, device=dev) < 0.5).float()
        sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()])
        opt = zs.optim.FlatAdam([list(model.generator.parameters()), list(model.variational.parameters())], lr=1e-3)
        return model, {"x": x}, sb, opt, zs.DeviceRNG(dev, seed=9)

    def stages(model, obs, sb, opt, rng, held):
        def s1():
            rng.begin_step()
            sb.zero()
            held["loss"] = model(obs)
            sb.backward_stage(held["loss"], 0)
            return held["loss"].detach()

        def s2():
            sb.backward_stage(held["loss"], 1)

        def s3():
            sb.scale(gradients=False)
            opt.step(grad_scale=sb.grad_scale())
        return [("graph", s1), ("eager", lambda: sb.launch(0)), ("graph", s2), ("eager", lambda: (sb.launch(1), sb.wait())),
                ("graph", s3)]
    me, oe, sbe, ope, re_ = make()
    mg, og, sbg, opg, rg = make()
    gs = zs.GraphedStages(stages(mg, og, sbg, opg, rg, {}), rng=rg, warmup=3, restore=True, optimizer=opg)
    for pe, pg in zip(me.parameters(), mg.parameters()):
        assert torch.equal(pe, pg), "restore=True must undo the warm-up passes"
    assert torch.equal(rg.state, re_.state)
    eager = stages(me, oe, sbe, ope, re_, {})
    le, lg = [], []
    with zs.device_rng(re_):
        for _ in range(4):
            first = None
            for kind, fn in eager:
                out = fn()
                if kind == "graph" and first is None:
                    first = out
            le.append(float(first))
    for _ in range(4):
        lg.append(float(gs()))
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    for pe, pg in zip(me.parameters(), mg.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["iwae", "vae"])
def test_graphed_training_over_a_stream_of_minibatches(kind):
    """The documented recipe for feeding a NEW minibatch to every replay (GraphedStep(inputs=...), step(x=batch)): graphed
    training over ten different batches equals eager training over the same stream (reference loop: iwae.py:151-160)."""
    dev = torch.device("cuda:0")
    model_e, opt_e, rng_e, obs_e = _make(kind, dev)
    model_g, opt_g, rng_g, obs_g = _make(kind, dev)
    B = obs_e["x"].shape[0]
    gen = torch.Generator().manual_seed(4)
    stream = [(torch.rand(B, 784, generator=gen) < 0.5).float() for _ in range(10)]          # host tensors, as a loader yields them
    step = zs.GraphedStep(_compute(model_g, rng_g, obs_g), opt_g.step, rng=rng_g, warmup=3, restore=True, inputs=obs_g)
    le, lg = [], []
    with zs.device_rng(rng_e):
        for xb in stream:
            le.append(float(_compute(model_e, rng_e, {"x": xb.to(dev)})()))
            opt_e.step()
    for xb in stream:
        lg.append(float(step(x=xb)))
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    assert len(set(np.round(lg, 3))) == 10
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)
    with pytest.raises(ValueError, match="shape"):
        step(x=torch.zeros(B + 1, 784))
    with pytest.raises(KeyError):
        step(y=torch.zeros(B))


@pytest.mark.gpu
def test_scratch_growth_after_a_capture_keeps_the_captured_set_alive():
    """A graph captured with one scratch set (workspaces + zero-initialised tickets of the hand-off kernels) keeps replaying
    correctly after an EAGER call has made the package allocate a bigger set (more particles than ticket words): the old set
    must stay allocated -- its pointers are baked into the graph."""
    from zhusuan import _ops
    dev = torch.device("cuda:0")
    model_e, opt_e, rng_e, obs = _make_bnn(dev)
    model_g, opt_g, rng_g, _ = _make_bnn(dev)
    step = zs.GraphedStep(_compute(model_g, rng_g, obs), opt_g.step, rng=rng_g, warmup=3, restore=True)
    comp_e = _compute(model_e, rng_e, obs)
    le, lg = [], []
    with zs.device_rng(rng_e):
        for _ in range(3):
            le.append(float(comp_e()))
            opt_e.step()
    for _ in range(3):
        lg.append(float(step()))
    # grow the per-particle ticket set eagerly (K = 200 > the 64 words the capture set was made with), then churn the allocator
    retired = len(_ops._SCRATCH_RETIRED)
    h = torch.randn(200, 8, 13, device=dev, requires_grad=True)
    w = torch.randn(200, 5, 14, device=dev, requires_grad=True)
    zs.particle_linear(h, w, relu=True).sum().backward()
    assert len(_ops._SCRATCH_RETIRED) > retired, "the replaced capture set must be kept, not freed"
    junk = [torch.full((1 << 16,), 7.0, device=dev) for _ in range(64)]
    torch.cuda.synchronize()
    del junk
    with zs.device_rng(rng_e):
        for _ in range(4):
            le.append(float(comp_e()))
            opt_e.step()
    for _ in range(4):
        lg.append(float(step()))
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["iwae", "vae"])
def test_replays_and_eager_steps_do_not_grow_the_allocator(kind):
    """A replay allocates nothing; eager steps return to their own baseline once the last loss is dropped (tools/soak.py
    asserts the same over 20 000 replays at the config shapes)."""
    import gc
    dev = torch.device("cuda:0")
    model, opt, rng, obs = _make(kind, dev)
    compute = _compute(model, rng, obs)
    step = zs.GraphedStep(compute, opt.step, rng=rng, warmup=3)

    def settled():
        gc.collect()
        torch.cuda.synchronize()
        return torch.cuda.memory_allocated()
    for _ in range(5):
        step()
    m5 = settled()
    for _ in range(60):
        step()
    assert settled() == m5
    with zs.device_rng(rng):
        for n, keep in ((3, "a"), (40, "b")):
            for _ in range(n):
                loss = compute()
                opt.step()
            del loss
            for p in model.parameters():
                p.grad = None
            if keep == "a":
                base = settled()
            else:
                assert settled() <= base


@pytest.mark.gpu
@pytest.mark.parametrize("which_fails", [None, 0, 2])
def test_a_failed_capture_with_agree_runs_the_same_stages_eagerly(which_fails):
    """zhusuan.GraphedStages(agree=...) / GraphedStep(agree=...) (several ranks record the same step): `agree` is called once
    after EVERY capture attempt, whatever happened; when it returns False (this rank's capture failed, or a peer's did) the
    object gives its graphs up and runs the same stages eagerly -- same values as eager training, same eager stages in the
    same order (the collectives of a multi-rank step), and no exception leaves the constructor."""
    from zhusuan import dataparallel
    dev = torch.device("cuda:0")

    def make():
        torch.manual_seed(3)
        model = vae_mnist.build(32, hidden=64, device=dev, dense="zhusuan")
        x = (torch.rand(32, 784, device=dev) < 0.5).float()
        sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()])
        opt = zs.optim.FlatAdam([list(model.generator.parameters()), list(model.variational.parameters())], lr=1e-3)
        return model, {"x": x}, sb, opt, zs.DeviceRNG(dev, seed=9)

    def stages(model, obs, sb, opt, rng, held, log, fail_at=None):
        def boom(i):
            if fail_at == i and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("injected capture failure in graph stage %d" % i)

        def s1():
            boom(0)
            rng.begin_step()
            sb.zero()
            held["loss"] = model(obs)
            sb.backward_stage(held["loss"], 0)
            return held["loss"].detach()

        def s2():
            boom(1)
            sb.backward_stage(held["loss"], 1)

        def s3():
            boom(2)
            sb.scale(gradients=False)
            opt.step(grad_scale=sb.grad_scale())
        return [("graph", s1), ("eager", lambda: (log.append("c0"), sb.launch(0))), ("graph", s2),
                ("eager", lambda: (log.append("c1"), sb.launch(1), sb.wait())), ("graph", s3)]
    me, oe, sbe, ope, re_ = make()
    mg, og, sbg, opg, rg = make()
    votes, log_g, log_e = [], [], []

    def agree(ok):
        votes.append(ok)
        return ok
    gs = zs.GraphedStages(stages(mg, og, sbg, opg, rg, {}, log_g, fail_at=which_fails), rng=rg, warmup=3, restore=True,
                          optimizer=opg, agree=agree)
    if which_fails is None:
        assert gs.captured and votes == [True, True, True] and len(gs.graphs) == 3
    else:
        assert not gs.captured and "injected capture failure" in gs.capture_error and gs.graphs == []
        assert votes == [True] * (which_fails if which_fails < 2 else 2) + [False]      # nobody votes after the job fell back
    del log_g[:]
    eager = stages(me, oe, sbe, ope, re_, {}, log_e)
    le, lg = [], []
    with zs.device_rng(re_):
        for _ in range(4):
            first = None
            for kind, fn in eager:
                out = fn()
                if kind == "graph" and first is None:
                    first = out
            le.append(float(first))
    for _ in range(4):
        lg.append(float(gs()))
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    assert log_g == log_e == ["c0", "c1"] * 4                  # the eager stages (collectives) ran once per step, in order
    for pe, pg in zip(me.parameters(), mg.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("peer_failed", [False, True])
def test_graphed_step_with_agree_falls_back_when_any_rank_says_so(peer_failed):
    dev = torch.device("cuda:0")
    model_e, opt_e, rng_e, obs = _make("iwae", dev)
    model_g, opt_g, rng_g, _ = _make("iwae", dev)
    calls = []

    def exchange(loss):
        calls.append(1)
        return loss
    comp = _compute(model_g, rng_g, obs)

    def compute():
        if not peer_failed and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected capture failure")
        return comp()
    votes = []

    def agree(ok):
        votes.append(ok)
        return ok and not peer_failed
    step = zs.GraphedStep(compute, opt_g.step, exchange=exchange, rng=rng_g, warmup=3, restore=True, agree=agree)
    assert votes == [peer_failed] and not step.captured and step.graphs == []
    assert ("injected" in step.capture_error) == (not peer_failed)
    calls.clear()
    comp_e = _compute(model_e, rng_e, obs)
    le, lg = [], []
    with zs.device_rng(rng_e):
        for _ in range(5):
            le.append(float(comp_e()))
            opt_e.step()
    for _ in range(5):
        lg.append(float(step()))
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    assert len(calls) == 5
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        np.testing.assert_allclose(pg.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)
