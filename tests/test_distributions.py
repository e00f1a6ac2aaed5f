"""Normal / Bernoulli through the product package.

Restates the reference's own unit tests (test/distributions/test_normal.py, test_bernoulli.py and the
shape tables of test/distributions/utils.py) and checks values + gradients against the golden
fixtures.  Every test runs on the "host" back-end (C oracle injected, CPU) and -- marked gpu -- on
the real HIP library.
"""
import numpy as np
import pytest
import torch
from scipy import stats

from conftest import load_golden
import host_backend
import zhusuan as zs
from zhusuan.distributions import Normal, Bernoulli


def T(a, dev, rg=False):
    x = torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    return x.requires_grad_(rg)


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ Normal: reference test_normal.py
def test_normal_init_errors(dev):
    # test_normal.py:29-43
    with pytest.raises(ValueError, match=r"Either.*should be passed"):
        Normal(mean=torch.zeros([2, 1]), std=torch.ones([2, 4, 3]), logstd=torch.zeros([2, 2, 3]), device=dev)
    with pytest.raises(ValueError, match=r"Either.*should be passed"):
        Normal(mean=torch.zeros([2, 1]), device=dev)
    with pytest.raises(RuntimeError):
        Normal(mean=torch.zeros([2, 1]), logstd=torch.zeros([2, 4, 3]), device=dev)
    with pytest.raises(TypeError, match="must have a dtype in"):
        Normal(mean=torch.zeros([2], dtype=torch.int32), std=torch.ones([2], dtype=torch.int32), device=dev)
    with pytest.raises(TypeError, match="must have the same dtype as"):
        Normal(mean=torch.zeros([2]), std=torch.ones([2], dtype=torch.float64), device=dev)
    with pytest.raises(ValueError, match="non-negative"):
        Normal(mean=0., std=1., group_ndims=-1, device=dev)
    d = Normal(mean=torch.ones([32, 1]), std=torch.ones([32, 1, 3]), device=dev)
    assert d.dtype == torch.float32
    assert Normal(mean=0., std=1., device=dev).dtype == torch.float32


def test_normal_sample_shapes(dev):
    # test/distributions/utils.py:286-303
    for ctor in ("std", "logstd"):
        for ms, ss, n, target in [([2, 3], [2, 1], 1, [2, 3]), ([1, 3], [2, 1], 2, [2, 2, 3]),
                                  ([2, 1, 5], [1, 3, 1], 3, [3, 2, 3, 5])]:
            kw = {ctor: torch.ones(ss) if ctor == "std" else torch.zeros(ss)}
            d = Normal(mean=torch.zeros(ms), device=dev, **kw)
            assert list(d.sample(n).shape) == target
    d = Normal(mean=torch.zeros([4, 5]), std=torch.ones([4, 5]), device=dev)
    assert list(d.sample().shape) == [4, 5]
    assert list(d.sample(None).shape) == [4, 5]
    assert list(d.sample(1).shape) == [4, 5]       # base.py:146-150: n_samples=1 adds no axis
    assert list(d.sample(7).shape) == [7, 4, 5]


def test_normal_batch_shape(dev):
    # test/distributions/utils.py:322-337
    for ms, ss, target in [([2, 3], [3], [2, 3]), ([2, 1, 4], [2, 3, 4], [2, 3, 4]), ([2, 3, 5], [3, 1], [2, 3, 5])]:
        assert list(Normal(mean=torch.zeros(ms), std=torch.ones(ss), device=dev).batch_shape) == target
    with pytest.raises(RuntimeError):
        Normal(mean=torch.zeros([2, 3, 5]), std=torch.ones([3, 2]), device=dev)


def test_normal_log_prob_shapes(dev):
    # test/distributions/utils.py:252-269
    for ms, ss, gs, target in [([2, 3], [2, 1], [1, 3], [2, 3]), ([1, 3], [1, 1], [2, 1, 3], [2, 1, 3]),
                               ([1, 5], [3, 1], [1, 2, 1, 1], [1, 2, 3, 5])]:
        d = Normal(mean=torch.zeros(ms), std=torch.ones(ss), device=dev)
        assert list(d.log_prob(torch.zeros(gs)).shape) == target
    # docs/tutorials/concepts.rst:75-79
    d = Normal(mean=torch.zeros([2, 1, 3]), std=1., group_ndims=2, device=dev)
    assert list(d.log_prob(torch.zeros([5, 1, 1, 3])).shape) == [5, 2]


def test_normal_property(dev):
    # test_normal.py:55-63
    mean, std = T([1., 2.], dev), T([1., 4.], dev)
    d = Normal(mean=mean, std=std)
    assert mean.equal(d.mean) and std.equal(d.std) and torch.log(std).equal(d.logstd)
    s = d.sample()
    assert torch.norm(torch.log(d.prob(s)) - d.log_prob(s)) < 1e-6


def test_normal_reparameterized_gradients(dev):
    # test_normal.py:65-84
    mean = torch.ones([2, 3], device=dev, requires_grad=True)
    logstd = torch.ones([2, 3], device=dev, requires_grad=True)
    s = Normal(mean=mean, logstd=logstd).sample()
    gm, gl = torch.autograd.grad(s.sum(), [mean, logstd], allow_unused=True)
    assert gm is not None and gl is not None
    close(gm, np.ones([2, 3]))
    s = Normal(mean=mean, logstd=logstd, is_reparameterized=False).sample()
    gm, gl = torch.autograd.grad(s.sum(), [mean, logstd], allow_unused=True)      # test_normal.py:77-84
    assert float(gm.abs().sum()) == 0.0 and float(gl.abs().sum()) == 0.0        # zero gradient, normal.py:102


def test_normal_known_values(dev):
    # test_normal.py:92-126, scipy logpdf, rtol 1e-3 there; tighter here
    def check(given, mean, logstd):
        mean, given, logstd = (np.array(v, np.float32) for v in (mean, given, logstd))
        target = np.array(stats.norm.logpdf(given, mean, np.exp(logstd)), np.float32)
        lp1 = Normal(mean=T(mean, dev), logstd=T(logstd, dev)).log_prob(T(given, dev))
        lp2 = Normal(mean=T(mean, dev), std=T(np.exp(logstd), dev)).log_prob(T(given, dev))
        close(lp1, target, 2e-5, 2e-6)
        close(lp2, target, 2e-5, 2e-6)
    check([0.], [0.], [0.])
    check([0.99, 0.9, 9., 99.], [1.], [-3., -1., 1., 10.])
    check([7.], [0., 4.], [[1., 2.], [3., 5.]])
    lp = Normal(mean=T([1.], dev), logstd=T([-3., -1., 1., 10.], dev)).log_prob(T([0.99, 0.9, 9., 99.], dev))
    close(lp, [2.06088996, 0.04411618, -6.24966812, -10.91894817], 2e-6, 2e-6)
    d = Normal(mean=T([[-1., 1.], [0., -2.]], dev), std=1., group_ndims=1, device=dev)
    close(d.log_prob(torch.zeros([1])), [-2.83787704, -3.83787727], 1e-6, 1e-6)   # concepts.rst:70-73


def test_float64_parameters(dev):
    # test/distributions/utils.py:test_dtype_2parameter / test_float_dtype_1parameter_discrete: float64 in,
    # float64 samples and log-probs out; values against scipy at double precision
    for dt in (torch.float32, torch.float64):
        d = Normal(mean=torch.tensor([0.01], dtype=dt), std=torch.ones([1], dtype=dt), device=dev)
        assert d.dtype == dt and d.sample(1).dtype == dt and d.sample().dtype == dt
        assert d.log_prob(torch.tensor([0.01], dtype=dt)).dtype == dt
        b = Bernoulli(torch.tensor([[2., 3.], [4., 5.]], dtype=dt), dtype=dt, device=dev)
        assert b.sample(2).dtype == dt and b.log_prob(torch.ones(2, 2, dtype=dt)).dtype == dt
    rng = np.random.RandomState(9)
    mu, ls, x = rng.standard_normal((3, 6, 5)), 0.3 * rng.standard_normal((6, 5)), 2 * rng.standard_normal((4, 3, 6, 5))
    mu_t = torch.tensor(mu, device=dev, requires_grad=True)
    sd_t = torch.tensor(np.exp(ls), device=dev, requires_grad=True)
    x_t = torch.tensor(x, device=dev, requires_grad=True)
    lp = Normal(mean=mu_t, std=sd_t, group_ndims=2).log_prob(x_t)
    ref = stats.norm.logpdf(x, mu, np.exp(ls)).sum((-1, -2))
    assert lp.dtype == torch.float64 and lp.shape == (4, 3)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), ref, rtol=1e-12, atol=1e-12)
    w = torch.tensor(rng.standard_normal((4, 3)), device=dev)
    gmu, gsd, gx = torch.autograd.grad((lp * w).sum(), [mu_t, sd_t, x_t])
    mu_c, sd_c, x_c = (torch.tensor(v, requires_grad=True) for v in (mu, np.exp(ls), x))
    ref_lp = (-0.5 * np.log(2 * np.pi) - torch.log(sd_c) - 0.5 * ((x_c - mu_c) / sd_c) ** 2).sum((-1, -2))
    rmu, rsd, rx = torch.autograd.grad((ref_lp * w.cpu()).sum(), [mu_c, sd_c, x_c])
    for a, b in ((gmu, rmu), (gsd, rsd), (gx, rx)):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-10, atol=1e-10)
    # fused sample + log-prob in double, explicit epsilon: z exact, log q at double accuracy
    eps = rng.standard_normal((7, 6, 5))
    dq = Normal(mean=torch.tensor(mu[0], device=dev), std=torch.tensor(np.exp(ls), device=dev), group_ndims=1)
    with zs.inject_epsilon([eps]):
        z = dq.sample(7)
    assert z.dtype == torch.float64
    assert np.array_equal(z.cpu().numpy(), mu[0] + np.exp(ls) * eps)
    np.testing.assert_allclose(dq.log_prob(None).cpu().numpy(), stats.norm.logpdf(z.cpu().numpy(), mu[0], np.exp(ls)).sum(-1),
                               rtol=1e-12, atol=1e-12)
    zf = Normal(mean=torch.zeros(4000, 4, dtype=torch.float64, device=dev),
                std=torch.ones(4000, 4, dtype=torch.float64, device=dev)).sample(3)
    assert zf.dtype == torch.float64 and abs(float(zf.mean())) < 0.03 and abs(float(zf.std()) - 1) < 0.03
    # Bernoulli and the IW reduction in double
    p = rng.uniform(0.01, 0.99, (3, 4, 16))
    xb = (rng.uniform(size=(4, 16)) < 0.5).astype(np.float64)
    lpb = Bernoulli(probs=torch.tensor(p, device=dev), group_ndims=1).log_prob(torch.tensor(xb, device=dev))
    refb = (xb * np.log(p + 1e-8) + (1 - xb) * np.log(1 - p + 1e-8)).sum(-1)
    np.testing.assert_allclose(lpb.cpu().numpy(), refb, rtol=1e-12, atol=1e-12)
    from zhusuan.variational.importance_weighted_objective import ImportanceWeightedObjective
    lpq = torch.tensor(-50 + rng.standard_normal((9, 5)), device=dev)
    lpp = torch.tensor(-550 + 5 * rng.standard_normal((9, 5)), device=dev)
    obj = ImportanceWeightedObjective(None, None, axis=0, estimator="vimco")
    cost = obj.vimco(lpp, lpq)
    assert cost.dtype == torch.float64
    lw = (lpp - lpq).cpu()
    np.testing.assert_allclose(obj.last_iw_bound.cpu().numpy(), (torch.logsumexp(lw, 0) - np.log(9)).numpy(), rtol=1e-12)
    np.testing.assert_allclose(zs.log_mean_exp(lpp, 0).cpu().numpy(), (torch.logsumexp(lpp.cpu(), 0) - np.log(9)).numpy(), rtol=1e-12)


# ------------------------------------------------------------------ Normal: golden fixtures
def test_normal_sample_logprob_golden(dev):
    g = load_golden("g_normal_sample")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        K = int(g[p + "K"])
        K = None if K < 0 else K
        mu, ls = T(g[p + "mu"], dev, True), T(g[p + "ls"], dev, True)
        kw = dict(is_reparameterized=bool(g[p + "reparam"]), group_ndims=int(g[p + "g"]))
        exact = True
        if int(g[p + "use_logstd"]):
            d = Normal(mean=mu, logstd=ls, **kw)     # std = exp(logstd) formed INSIDE the kernel (sigma_is_logstd):
            exact = False                            # its exp differs from torch-CPU's vectorised exp in the last ulp
        else:
            # std = exp(ls) exactly as the reference run had it (stored: a CPU's vectorised exp can differ
            # in the last ulp from another CPU's), with the chain rule back to ls applied by hand below
            sd = T(g[p + "sd"], dev, True)
            d = Normal(mean=mu, std=sd, **kw)
        with zs.inject_epsilon([g[p + "eps"]]):
            z = d.sample(K)
        assert tuple(z.shape) == g[p + "z"].shape
        if exact:
            assert np.array_equal(z.detach().cpu().numpy(), g[p + "z"]), "z must be bit-exact (case %d)" % c
        else:
            close(z, g[p + "z"], 2e-6, 2e-6)
        lp = d.log_prob(None)
        close(lp, g[p + "lp"], 1e-5, 2e-5)
        obj = (lp * T(g[p + "w"], dev)).sum() + (z * T(g[p + "wz"], dev)).sum()
        if int(g[p + "use_logstd"]):
            gmu, gls = torch.autograd.grad(obj, [mu, ls], allow_unused=True)
        else:
            gmu, gsd = torch.autograd.grad(obj, [mu, sd], allow_unused=True)
            gls = None if gsd is None else gsd * sd.detach()          # d sd / d ls = sd
        gmu = gmu if gmu is not None else torch.zeros_like(mu)
        gls = gls if gls is not None else torch.zeros_like(ls)
        close(gmu, g[p + "gmu"], 1e-4, 1e-4)
        close(gls, g[p + "gls"], 1e-4, 2e-4)


def test_normal_logprob_given_golden(dev):
    g = load_golden("g_normal_logprob")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        mu, sd, x = T(g[p + "mu"], dev, True), T(g[p + "sd"], dev, True), T(g[p + "x"], dev, True)
        lp = Normal(mean=mu, std=sd, group_ndims=int(g[p + "g"])).log_prob(x)
        assert tuple(lp.shape) == g[p + "lp"].shape
        close(lp, g[p + "lp"], 1e-5, 1e-5)
        gmu, gsd, gx = torch.autograd.grad((lp * T(g[p + "w"], dev)).sum(), [mu, sd, x])
        close(gmu, g[p + "gmu"], 1e-4, 1e-4)
        close(gsd, g[p + "gsd"], 1e-4, 1e-4)
        close(gx, g[p + "gx"], 1e-4, 1e-4)


def test_normal_eps_shared_along_std_only_axes(dev):
    # normal.py:91-92,104: the draw has mean's shape
    g = load_golden("g_normal_epsshape")
    d = Normal(mean=T(g["mu"], dev), std=T(g["sd"], dev))
    with zs.inject_epsilon([g["eps"]]):
        z = d.sample(2)
    assert np.array_equal(z.cpu().numpy(), g["z"])
    close(d.log_prob(None), g["lp"], 1e-5, 1e-6)
    z2 = d.sample(2)                       # Philox path: rows along the std-only axis share their draw
    e = (z2 - d.mean) / d.std
    close(e[:, 0], e[:, 1], 1e-4, 1e-5)


def test_normal_philox_statistics_and_reproducibility(dev):
    from zhusuan import _rng
    n = 1 << 18
    mu = torch.full([n // 4, 4], 1.5, device=dev)
    sd = torch.full([n // 4, 4], 0.5, device=dev)
    d = Normal(mean=mu, std=sd, group_ndims=1)
    torch.manual_seed(7)
    host_backend.manual_seed(7)
    z = d.sample(2)
    lp = d.log_prob(None)
    e = ((z - 1.5) / 0.5).double().cpu().numpy().ravel()
    assert abs(e.mean()) < 6e-3 and abs(e.std() - 1) < 6e-3
    assert stats.kstest(e[:50000], "norm").pvalue > 1e-4
    assert abs(stats.skew(e)) < 0.02 and abs(stats.kurtosis(e)) < 0.05
    ref = stats.norm.logpdf(z.double().cpu().numpy(), 1.5, 0.5).sum(-1)
    close(lp, ref, 1e-5, 1e-4)
    torch.manual_seed(7)
    host_backend.manual_seed(7)
    z_again = d.sample(2)
    assert torch.equal(z, z_again)
    z_next = d.sample(2)
    assert not torch.equal(z, z_next)


# ------------------------------------------------------------------ Bernoulli: reference test_bernoulli.py
def test_bernoulli_init(dev):
    # test_bernoulli.py:20-33
    ber = Bernoulli(0., device=dev)
    assert ber.dtype == torch.float32
    assert float(ber.probs) == 0.5
    ber = Bernoulli(probs=[0.4, 0.5], device=dev)
    p = ber.probs
    assert ber.logits.equal(torch.log(p / (torch.ones_like(p) - p)))
    with pytest.raises(ValueError, match=r"Either.*should be passed"):
        Bernoulli(logits=1, probs=0.1, device=dev)
    with pytest.raises(ValueError, match=r"Either.*should be passed"):
        Bernoulli(device=dev)
    with pytest.raises(TypeError, match=r"must have a dtype in"):
        Bernoulli(probs=0, dtype=torch.int64, device=dev)
    for bad in (torch.int16, torch.float16, torch.uint8, torch.bool):
        with pytest.raises(TypeError):
            Bernoulli(torch.tensor([1.]), dtype=bad, device=dev)
    assert Bernoulli(logits=torch.zeros(2), device=dev).is_reparameterized is False


def test_bernoulli_property_and_shapes(dev):
    logits = torch.rand([2, 2], device=dev)
    ber = Bernoulli(logits=logits)
    assert logits.equal(ber.logits)
    close(ber.probs, torch.sigmoid(logits), 1e-6, 1e-7)
    s = ber.sample()
    assert set(np.unique(s.cpu().numpy())) <= {0.0, 1.0}
    assert torch.norm(torch.log(ber.prob(s)) - ber.log_prob(s)) < 1e-5
    for shp in ([], [2], [2, 3], [2, 1, 4]):
        assert list(Bernoulli(torch.ones(shp), device=dev).batch_shape) == shp
    for shp, n, target in [([2, 3], 1, [2, 3]), ([1, 3], 2, [2, 1, 3]), ([2, 1, 5], 3, [3, 2, 1, 5])]:
        s = Bernoulli(torch.ones(shp), device=dev).sample(n)
        assert list(s.shape) == target and s.dtype == torch.float32
    for ps, gs, target in [([2, 3], [1, 3], [2, 3]), ([1, 3], [2, 2, 3], [2, 2, 3]), ([1, 5], [1, 2, 3, 1], [1, 2, 3, 5])]:
        assert list(Bernoulli(torch.ones(ps), device=dev).log_prob(torch.ones(gs)).shape) == target


def test_bernoulli_known_values(dev):
    # test_bernoulli.py:56-73
    for logits, given in [([0.], [0.]), ([2., 1.], [0., 1.])]:
        logits = np.array(logits, np.float32)
        prob = 1. / (1. + np.exp(-logits))
        given = np.array(given, np.float32)
        target = stats.bernoulli.logpmf(given, prob)
        close(Bernoulli(logits, device=dev).log_prob(given), target, 1e-5, 1e-6)
        close(Bernoulli(probs=prob, device=dev).log_prob(given), target, 1e-5, 1e-6)
    close(Bernoulli(logits=T([2., 1.], dev)).log_prob(T([0., 1.], dev)), [-2.12692761, -0.31326166], 2e-6, 2e-6)
    # +1e-8 is part of the contract: p in {0, 1}
    close(Bernoulli(probs=T([0., 1.], dev)).log_prob(T([1., 1.], dev)), [-18.420681, 0.0], 1e-6, 1e-6)


def test_bernoulli_golden(dev):
    g = load_golden("g_bernoulli")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        gnd = int(g[p + "g"])
        x = T(g[p + "x"], dev)
        if int(g[p + "from_logits"]):
            lg = T(g[p + "logits"], dev, True)
            d = Bernoulli(logits=lg, group_ndims=gnd)
            close(d.probs, g[p + "probs"], 1e-6, 1e-7)
            lp = d.log_prob(x)
            close(lp, g[p + "lp"], 1e-5, 2e-6)
            (gl,) = torch.autograd.grad((lp * T(g[p + "w"], dev)).sum(), [lg])
            close(gl, g[p + "gl"], 1e-4, 1e-5)
        else:
            pr = T(g[p + "probs"], dev, True)
            lp = Bernoulli(probs=pr, group_ndims=gnd).log_prob(x)
            assert tuple(lp.shape) == g[p + "lp"].shape
            close(lp, g[p + "lp"], 1e-5, 1e-4 if gnd else 2e-6)
            if (p + "gp") in g.files:
                (gp,) = torch.autograd.grad((lp * T(g[p + "w"], dev)).sum(), [pr])
                close(gp, g[p + "gp"], 1e-4, 1e-4)


def test_bernoulli_sample_rate(dev):
    p = torch.tensor([0.1, 0.5, 0.9], device=dev).repeat(4)
    s = Bernoulli(probs=p).sample(20000)
    rate = s.mean(0).cpu().numpy()
    np.testing.assert_allclose(rate, p.cpu().numpy(), atol=0.012)


def test_cpu_tensor_without_gpu_library_fails_loudly():
    # no hook installed: a CPU tensor must not be silently computed somewhere else
    assert not host_backend.active()
    d = Normal(mean=torch.zeros(4), std=torch.ones(4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        d.sample()
    with pytest.raises(RuntimeError, match="no CPU path"):
        Bernoulli(probs=torch.full([4], 0.5)).log_prob(torch.ones(4))


def test_non_reparameterised_normal_draws_one_epsilon_per_broadcast_element(dev):
    """normal.py:102: the non-reparameterised draw is torch.normal(mean_rep, std_rep) -- one independent value per element of
    the BROADCAST shape (the reparameterised draw has mean's shape and is shared along std-only axes, :104)."""
    mean = torch.zeros(1, 3, device=dev)
    std = torch.ones(2, 1, device=dev)
    d = Normal(mean=mean, std=std, is_reparameterized=False)
    z = d.sample(4)
    assert tuple(z.shape) == (4, 2, 3) and not z.requires_grad
    assert not torch.equal(z[:, 0], z[:, 1])                    # rows along the std-only axis are independent draws
    eps = np.random.RandomState(5).standard_normal((4, 2, 3)).astype(np.float32)
    with zs.inject_epsilon([eps]):                              # injected epsilon has the broadcast shape too
        z2 = d.sample(4)
    close(z2, eps, 1e-6, 1e-6)
    close(d.log_prob(None), stats.norm.logpdf(eps), 1e-5, 1e-5)
    dr = Normal(mean=mean, std=std)                             # reparameterised: shared along the std-only axis
    zr = dr.sample(4)
    close(zr[:, 0], zr[:, 1], 1e-6, 1e-6)


# ------------------------------------------------------------------ device-resident RNG state (hipGraph-safe draws)
def test_device_rng_state_drives_the_draws(dev):
    mu = torch.zeros([64, 8], device=dev)
    sd = torch.ones([64, 8], device=dev)
    rng = zs.DeviceRNG(dev, seed=5)
    with zs.device_rng(rng):
        rng.begin_step()
        a1 = Normal(mean=mu, std=sd).sample(3)
        a2 = Normal(mean=mu, std=sd).sample(3)          # next delta inside the same step
        rng.begin_step()
        b1 = Normal(mean=mu, std=sd).sample(3)
    assert not torch.equal(a1, a2) and not torch.equal(a1, b1)
    rng2 = zs.DeviceRNG(dev, seed=5)
    with zs.device_rng(rng2):
        rng2.begin_step()
        c1 = Normal(mean=mu, std=sd).sample(3)
    assert torch.equal(a1, c1)                           # same (seed, base, delta) -> same draw
    rng3 = zs.DeviceRNG(dev, seed=6)
    with zs.device_rng(rng3):
        rng3.begin_step()
        d1 = Normal(mean=mu, std=sd, is_reparameterized=True).sample(3)
    assert not torch.equal(a1, d1)
    # backward regenerates the same eps from the device state
    m = torch.zeros([64, 8], device=dev, requires_grad=True)
    s = torch.full([64, 8], 2.0, device=dev, requires_grad=True)
    rng4 = zs.DeviceRNG(dev, seed=5)
    with zs.device_rng(rng4):
        rng4.begin_step()
        z = Normal(mean=m, std=s).sample(3)
        (gs,) = torch.autograd.grad(z.sum(), [s])
    close(gs, ((z.detach() - 0.0) / 2.0).sum(0), 1e-5, 1e-5)


def test_device_rng_backward_after_begin_step(dev):
    """a backward that runs after DeviceRNG.begin_step() (two objectives per step, retain_graph, delayed
    backward) must regenerate the FORWARD's epsilon, not the one the advanced live state would give."""
    from zhusuan.distributions import Logistic
    m = torch.zeros([16, 8], device=dev, requires_grad=True)
    s = torch.full([16, 8], 1.5, device=dev, requires_grad=True)
    ls = torch.full([16, 8], 0.25, device=dev, requires_grad=True)
    rng = zs.DeviceRNG(dev, seed=11)
    with zs.device_rng(rng):
        rng.begin_step()
        z = Normal(mean=m, std=s).sample(4)
        zl = Normal(mean=m, logstd=ls).sample(4)
        sg = torch.full([16, 8], 0.5, device=dev, requires_grad=True)
        zg = Logistic(loc=m, scale=sg).sample(4)
        obj = (z * z).sum() + (zl * zl).sum() + (zg * zg).sum()
        before = torch.autograd.grad(obj, [s, ls, sg, m], retain_graph=True)
        rng.begin_step()
        Normal(mean=m, std=s).sample(4)                  # the next step's draws move the live state further
        after = torch.autograd.grad(obj, [s, ls, sg, m])
    for a_, b_ in zip(before, after):
        assert torch.equal(a_, b_)
    # and the gradient is the one of the draw that was made: d sum(z^2) / d sigma = sum_k 2 z eps, eps = (z - m) / sigma
    close(before[0], (2 * z.detach() * (z.detach() / 1.5)).sum(0), 1e-4, 1e-4)
    close(before[1], (2 * zl.detach() * zl.detach()).sum(0), 1e-4, 1e-4)     # d/d logstd = sigma * d/d sigma
    close(before[2], (2 * zg.detach() * (zg.detach() / 0.5)).sum(0), 1e-4, 1e-4)


@pytest.mark.gpu
def test_hipgraph_replay_draws_fresh_numbers():
    from examples import iwae
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = iwae.build(n_samples=5, estimator="vimco", hidden=32, device=dev)
    x = (torch.rand(16, 784, device=dev) < 0.5).float()
    opt = torch.optim.Adam(model.parameters(), 1e-3, fused=True, capturable=True)
    rng = zs.DeviceRNG(dev, seed=3)

    def body():
        rng.begin_step()
        opt.zero_grad(set_to_none=False)
        loss = model({"x": x})
        loss.backward()
        opt.step()
        return loss.detach()

    with zs.device_rng(rng):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = body()
        vals = []
        for _ in range(4):
            g.replay()
            vals.append(float(out))
    assert all(np.isfinite(v) for v in vals)
    assert len(set(vals)) == 4, vals                      # fresh epsilon on every replay
