"""IW1 (include/zs_hip.h: zs_bernoulli_iw_objective / _bwd): the generator side of the importance-weighted objective in one
launch -- Bernoulli row sums + the Normal prior's log-density of the latent + the sum of the generator's terms - log q + the
K-particle reductions + the batch mean (zhusuan/variational/importance_weighted_objective.py:66-191 over normal.py:109-126 and
bernoulli.py:84-95 of the reference).

not gpu : the C oracle's IW1 is bit for bit the composition of its K3, K2 and K4b (which the goldens pin); argument checks;
          the package's fused path equals its per-node path (host back-end).
gpu     : libzs_hip.so against the oracle and against its own unfused kernels on seeded inputs: every combination of the
          optional terms, scalar / repeated prior parameters, probabilities / logits, shared / full-size observations, ragged
          row counts, K = 2 .. 64, repeated launches on ONE scratch set with different numbers (a stale hand-off would show in
          the second), tickets back at zero, a deterministic batch mean; the backward with the incoming gradient as a device
          scalar / a per-datapoint vector.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import host_kernel_library
from test_cabi import Raw, orc, hip, orc64, hip64      # noqa: F401  (fixtures)
from zhusuan import _hip


def _inputs(rng, K, R, D, Dz, logits, x_full, pm_scalar, ps_scalar, ls):
    p = rng.uniform(-4, 4, size=(K, R, D)) if logits else rng.uniform(0.02, 0.98, size=(K, R, D))
    p.reshape(-1)[::97] = 4.0 if logits else 0.999999            # confident pixels: the + 1e-8 matters
    x = (rng.uniform(size=((K, R, D) if x_full else (R, D))) < 0.5).astype(np.float64)
    z = rng.standard_normal((K, R, Dz))
    pmu = rng.standard_normal(1 if pm_scalar else (R, Dz)) * 0.3
    psg = rng.uniform(0.5, 1.5, size=(1 if ps_scalar else (R, Dz)))
    if ls:
        psg = np.log(psg)
    rows_a = -20 + rng.standard_normal((R, K))
    logq = -45 + 2 * rng.standard_normal((R, K))
    return p, x, z, pmu, psg, rows_a, logq


class Scratch(object):
    """The accumulator word of the batch mean, reused across launches like the package's per-stream scratch."""

    def __init__(self, raw, R):
        self.acc = torch.zeros(64, dtype=torch.int64, device=raw.dev)


def iw1(raw, p, x, K, R, D, z, pmu, psg, ls, rows_a, logq, est, want_mean, logits, scratch=None):
    sc = scratch or Scratch(raw, R)
    Dz = z.shape[-1] if z is not None else 1
    lp_x, lp_z = raw.empty(R, K), raw.empty(R, K)
    cost_b, bound, coef, mean = raw.empty(R), raw.empty(R), raw.empty(2, R, K), raw.empty(1)
    raw.call("zs_bernoulli_iw_objective_f32", raw.t(p), int(logits), raw.t(x), int(x.size), K, R, D,
             raw.t(z), raw.t(pmu), int(pmu.size) if pmu is not None else 1, raw.t(psg), int(psg.size) if psg is not None else 1,
             Dz, int(ls), raw.t(rows_a), K, raw.t(logq), K, est, int(want_mean), lp_x, lp_z if z is not None else None,
             cost_b, bound, coef, mean if want_mean else None, sc.acc)
    assert int(sc.acc.abs().sum().item()) == 0, "the accumulator must be handed back at zero"
    return dict(lp_x=lp_x.cpu().numpy(), lp_z=lp_z.cpu().numpy() if z is not None else None, cost=cost_b.cpu().numpy(),
                bound=bound.cpu().numpy(), coef=coef.cpu().numpy(), mean=mean.cpu().numpy() if want_mean else None)


def composed(raw, p, x, K, R, D, z, pmu, psg, ls, rows_a, logq, est, want_mean, logits, lp_x=None, lp_z=None):
    """The same objective from the separate entry points: K3, K2, the additions on the host in fp32 (left to right), K4b."""
    dt = np.float32 if raw.dtype == torch.float32 else np.float64
    if lp_x is None:
        out = raw.empty(R, K)
        if logits:
            raw.call("zs_bernoulli_logits_logprob_f32", raw.t(p), raw.t(x), int(x.size), out, None, K, R, D, 1, K)
        else:
            raw.call("zs_bernoulli_logprob_f32", raw.t(p), raw.t(x), int(x.size), out, K, R, D, 1, K)
        lp_x = out.cpu().numpy()
    if z is not None and lp_z is None:
        out = raw.empty(R, K)
        raw.call("zs_normal_logprob_f32", raw.t(z), int(z.size), raw.t(pmu), int(pmu.size), raw.t(psg), int(psg.size), out, K, R,
                 z.shape[-1], 1, K, int(ls))
        lp_z = out.cpu().numpy()
    total = lp_x.astype(dt)
    if z is not None:
        head = lp_z.astype(dt) if rows_a is None else (rows_a.astype(dt) + lp_z.astype(dt)).astype(dt)
        total = (head + lp_x.astype(dt)).astype(dt)
    elif rows_a is not None:
        total = (rows_a.astype(dt) + lp_x.astype(dt)).astype(dt)
    cost_b, bound, coef, mean = raw.empty(R), raw.empty(R), raw.empty(2, R, K), raw.empty(1)
    ws, ticket = raw.empty(4096), torch.zeros(1, dtype=torch.int32, device=raw.dev)
    raw.call("zs_iw_objective_f32", raw.t(total), K, None, K, raw.t(logq), K, R, K, est, int(want_mean), cost_b, bound, coef,
             mean if want_mean else None, ws, 4096, ticket)
    return dict(lp_x=lp_x, lp_z=lp_z, cost=cost_b.cpu().numpy(), bound=bound.cpu().numpy(), coef=coef.cpu().numpy(),
                mean=mean.cpu().numpy() if want_mean else None)


CASES = [  # K, R, D, Dz, logits, x_full, pm_scalar, ps_scalar, ls, with_z, with_rows
    (5, 8, 784, 40, False, False, False, False, False, True, False),
    (50, 37, 784, 40, False, False, False, False, False, True, False),
    (2, 1, 256, 4, True, False, True, True, False, True, True),
    (64, 5, 1024, 13, False, True, False, True, True, True, True),
    (7, 130, 260, 3, True, False, True, False, False, True, False),
    (50, 64, 784, 40, True, False, False, False, True, False, True),
    (33, 3, 512, 8, False, False, False, False, False, False, False),
]


def test_c_oracle_iw1_is_the_composition(orc):
    rng = np.random.RandomState(7)
    for (K, R, D, Dz, logits, x_full, pms, pss, ls, with_z, with_rows) in CASES[:5] + CASES[5:]:
        R, D = min(R, 6), min(D, 256)              # the serial oracle: keep it small
        p, x, z, pmu, psg, rows_a, logq = _inputs(rng, K, R, D, Dz, logits, x_full, pms, pss, ls)
        f = lambda a: None if a is None else a.astype(np.float32)
        args = (f(p), f(x), K, R, D, f(z) if with_z else None, f(pmu) if with_z else None, f(psg) if with_z else None, ls,
                f(rows_a) if with_rows else None, f(logq))
        for est in (0, 1):
            for want_mean in (True, False):
                got = iw1(orc, *args, est, want_mean, logits)
                ref = composed(orc, *args, est, want_mean, logits)
                for key in ("lp_x", "cost", "bound", "coef") + (("lp_z",) if with_z else ()) + (("mean",) if want_mean else ()):
                    np.testing.assert_array_equal(got[key], ref[key], err_msg=key)


def test_c_oracle_iw1_rejects_bad_arguments(orc):
    rng = np.random.RandomState(1)
    K, R, D = 3, 2, 256
    p, x, z, pmu, psg, rows_a, logq = [None if a is None else a.astype(np.float32) for a in _inputs(rng, K, R, D, 4, False, False, False, False, False)]
    sc = Scratch(orc, R)
    ok = lambda: [orc.t(p), 0, orc.t(x), x.size, K, R, D, orc.t(z), orc.t(pmu), pmu.size, orc.t(psg), psg.size, 4, 0, None, K, orc.t(logq), K, 1, 1,
                  orc.empty(R, K), orc.empty(R, K), orc.empty(R), orc.empty(R), orc.empty(2, R, K), orc.empty(1), sc.acc]
    orc.call("zs_bernoulli_iw_objective_f32", *ok())
    for idx, bad in ((20, None), (22, None), (26, None), (18, 7), (9, 3), (25, None), (17, K - 1)):
        a = ok()
        a[idx] = bad
        with pytest.raises(RuntimeError, match="code -1"):
            orc.call("zs_bernoulli_iw_objective_f32", *a)
    a = ok()
    a[3] = D                                        # an observation period that is neither R*D nor K*R*D
    with pytest.raises(RuntimeError, match="code -2"):
        orc.call("zs_bernoulli_iw_objective_f32", *a)
    a = ok()
    a[4], a[18] = 1, 1                              # VIMCO with one particle
    with pytest.raises(RuntimeError, match="code -1"):
        orc.call("zs_bernoulli_iw_objective_f32", *a)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [(50, 256, 784, 40, False, False, False, False, False, True, False),
                                          (40, 1000, 784, 40, True, False, True, True, False, True, False),
                                          # the persistent grid (round 5): one workgroup per CU, several datapoints each -- one more than
                                          # the CUs, a ragged multiple, K = NW and K just above it (flat rows straddle datapoints every
                                          # round), two particles, the widest latent row, tens of thousands of datapoints
                                          (50, 257, 784, 40, False, False, False, False, False, True, False),
                                          (50, 512, 784, 40, False, False, False, False, False, True, True),
                                          (16, 773, 512, 8, True, False, False, False, True, True, False),
                                          (17, 600, 784, 40, False, False, True, False, False, True, False),
                                          (2, 3000, 256, 4, False, False, False, True, False, True, True),
                                          (64, 515, 1024, 256, False, False, False, False, False, True, False),
                                          (3, 40000, 256, 4, False, False, True, True, False, False, False),
                                          (9, 300, 640, 12, False, True, False, False, False, True, False)])
def test_hip_iw1_forward(hip, orc, case):
    (K, R, D, Dz, logits, x_full, pms, pss, ls, with_z, with_rows) = case
    rng = np.random.RandomState(K * 1000 + R + D)
    sc = Scratch(hip, R)                            # ONE scratch set for every launch of this test
    if with_z and Dz % 4 != 0:
        # a latent row that is not a whole number of 16-byte pieces is outside the fused kernel's domain: refused, not mis-read
        p, x, z, pmu, psg, rows_a, logq = [a.astype(np.float32) for a in _inputs(rng, K, R, D, Dz, logits, x_full, pms, pss, ls)]
        with pytest.raises(RuntimeError, match="code -2"):
            iw1(hip, p, x, K, R, D, z, pmu, psg, ls, None, logq, 0, True, logits, scratch=sc)
        with_z, with_rows = False, True             # ... and the caller's alternative: that node's rows from its own kernel
    for rep in range(2):
        p, x, z, pmu, psg, rows_a, logq = _inputs(rng, K, R, D, Dz, logits, x_full, pms, pss, ls)
        if rep == 1:                               # fractional observations: the two-logarithm form (rows of bits take the one-logarithm form)
            x = rng.uniform(size=x.shape)
            x[: max(R // 2, 1)].reshape(-1)[::5] = 1.0
        f = lambda a: None if a is None else a.astype(np.float32)
        args = (f(p), f(x), K, R, D, f(z) if with_z else None, f(pmu) if with_z else None, f(psg) if with_z else None, ls,
                f(rows_a) if with_rows else None, f(logq))
        for est in (0, 1):
            for want_mean in (True, False):
                got = iw1(hip, *args, est, want_mean, logits, scratch=sc)
                # (1) the unfused kernels of the same library (another order of summation inside a row: last-bit differences) ...
                base = composed(hip, *args, est, want_mean, logits)
                np.testing.assert_allclose(got["lp_x"], base["lp_x"], rtol=1e-6, atol=1e-4)
                if with_z:
                    np.testing.assert_allclose(got["lp_z"], base["lp_z"], rtol=1e-5, atol=1e-4)
                # ... and fed ITS OWN rows, K4b reproduces every output bit for bit (the tail is K4b's wave kernel)
                if K <= 64 and R < 4096:
                    same = composed(hip, *args, est, want_mean, logits, lp_x=got["lp_x"], lp_z=got["lp_z"])
                    for key in ("cost", "bound", "coef"):
                        np.testing.assert_array_equal(got[key], same[key], err_msg=key)
                assert np.isfinite(got["cost"]).all()
                if want_mean:                      # the batch mean: the mean of the fp32 costs, correctly rounded (two fixed-point words)
                    exact = got["cost"].astype(np.float64).mean()
                    assert abs(got["mean"][0] - exact) <= 0.51 * 2.0 ** -23 * abs(exact), (got["mean"][0], exact)
                # (2) the oracle
                if K * R * D <= 2_000_000:
                    ref = iw1(orc, *args, est, want_mean, logits)
                    np.testing.assert_allclose(got["lp_x"], ref["lp_x"], rtol=2e-5, atol=1e-3)
                    np.testing.assert_allclose(got["bound"], ref["bound"], rtol=2e-5, atol=1e-3)
                    if want_mean:
                        assert abs(got["mean"][0] - ref["mean"][0]) <= 5e-5 * abs(ref["mean"][0]) + 1e-3
                if want_mean:
                    again = iw1(hip, *args, est, True, logits, scratch=sc)
                    assert got["mean"][0] == again["mean"][0], "the batch mean must be deterministic"


@pytest.mark.gpu
@pytest.mark.parametrize("R", [1, 7, 256, 300, 4096, 32768])
def test_hip_iw1_batch_mean_of_small_and_mixed_costs(hip, R):
    """the batch mean is a fixed-point sum (deterministic: integer addition).  With ONE word per sum its
    resolution was absolute (2^-21 per datapoint at R = 256, 2^-11 at R = 32 768): costs of a converged toy model, 1e-4 .. 1e-6 with
    mixed signs, lost relative precision against the fp32 mean the reference takes
    (zhusuan/variational/importance_weighted_objective.py:191).  Two words (value + rounding residual): the mean of the fp32 costs,
    correctly rounded, at every magnitude; never further from the exact mean than numpy's fp32 mean."""
    K, D = 2, 256
    rng = np.random.RandomState(R)
    sc = Scratch(hip, R)
    for mag in (1e-4, 1e-5, 1e-6, 3.0):
        # log w = lp_x - log q with lp_x = D * log(p + 1e-8) ~ -D * delta (x = 1, p = 1 - delta) and log q of the same size, either sign
        delta = rng.uniform(0.2, 1.0, size=(K, R, 1)) * mag / D
        p = np.broadcast_to((1.0 - delta), (K, R, D)).astype(np.float32).copy()
        x = np.ones((R, D), np.float32)
        logq = (rng.uniform(-1.5, 0.5, size=(R, K)) * mag).astype(np.float32)
        for est in (0, 1):
            got = iw1(hip, p, x, K, R, D, None, None, None, False, None, logq, est, True, False, scratch=sc)
            cost = got["cost"].astype(np.float64)
            assert np.isfinite(cost).all() and (mag > 1 or np.abs(cost).max() < 50 * mag)
            if est == 0 and 5e-6 < mag < 1 and R >= 7:
                assert (cost > 0).any() and (cost < 0).any(), "the construction should give costs of both signs"
            exact = cost.mean()
            err = abs(float(got["mean"][0]) - exact)
            err32 = abs(float(got["cost"].mean(dtype=np.float32)) - exact)
            half_ulp = 0.5 * abs(exact) * 2.0 ** -23 + 1e-45
            assert err <= 1.01 * half_ulp + 1e-30, (mag, est, got["mean"][0], exact, err, half_ulp)
            assert err <= err32 + 1.01 * half_ulp
            again = iw1(hip, p, x, K, R, D, None, None, None, False, None, logq, est, True, False, scratch=sc)
            assert got["mean"][0] == again["mean"][0]


@pytest.mark.gpu
def test_hip_iw1_batch_mean_with_non_finite_costs(hip):
    """a +inf / -inf / NaN per-datapoint cost used to poison the fixed-point mean to NaN whatever it was; the
    fp32 mean the reference takes -- and K4b, the path beyond the fused domain -- return +inf / -inf / NaN.  Sticky flags per kind
    now reproduce that; the words are handed back at zero all the same, and the next launch is clean."""
    K, R, D = 4, 300, 256
    rng = np.random.RandomState(5)
    p, x, z, pmu, psg, rows_a, logq = [a.astype(np.float32) for a in _inputs(rng, K, R, D, 4, False, False, False, False, False)]
    sc = Scratch(hip, R)
    clean = iw1(hip, p, x, K, R, D, None, None, None, False, None, logq, 0, True, False, scratch=sc)
    assert np.isfinite(clean["mean"][0])
    cases = []
    lq = logq.copy(); lq[7, :] = np.inf                         # log w = -inf for every particle of datapoint 7: its bound is -inf ...
    cases.append((lq, None))
    lq = logq.copy(); lq[[3, 299], 1] = -np.inf                 # log w = +inf for one particle of two datapoints
    cases.append((lq, None))
    lq = logq.copy(); lq[11, 2] = np.nan
    cases.append((lq, None))
    lq = logq.copy(); lq[7, :] = np.inf; lq[200, 0] = -np.inf   # both signs
    cases.append((lq, None))
    for lq, _ in cases:
        for est in (0, 1):
            got = iw1(hip, p, x, K, R, D, None, None, None, False, None, lq, est, True, False, scratch=sc)
            with np.errstate(invalid="ignore"):
                want = got["cost"].astype(np.float32).mean(dtype=np.float32)          # what a float mean of these costs gives
            assert not np.isfinite(got["cost"]).all()
            if np.isnan(want):
                assert np.isnan(got["mean"][0]), (got["mean"][0], want)
            else:
                assert got["mean"][0] == want, (got["mean"][0], want)
            after = iw1(hip, p, x, K, R, D, None, None, None, False, None, logq, 0, True, False, scratch=sc)
            assert after["mean"][0] == clean["mean"][0]
    big = logq.copy(); big[5, :] = -3.0e7                       # a finite cost beyond 2^24: refused as NaN (documented in include/zs_hip.h)
    got = iw1(hip, p, x, K, R, D, None, None, None, False, None, big, 0, True, False, scratch=sc)
    assert np.isfinite(got["cost"]).all() and np.abs(got["cost"]).max() > 2.0 ** 24 and np.isnan(got["mean"][0])


@pytest.mark.gpu
def test_hip_iw1_outside_the_fused_domain(hip):
    rng = np.random.RandomState(3)
    for K, R, D in ((5, 4, 100), (65, 2, 256), (5, 4, 2048), (5, 4, 258)):
        p, x, z, pmu, psg, rows_a, logq = [a.astype(np.float32) for a in _inputs(rng, K, R, D, 4, False, False, False, False, False)]
        with pytest.raises(RuntimeError, match="code -2"):
            iw1(hip, p, x, K, R, D, None, None, None, False, None, logq, 0, True, False)


def iw1_bwd(raw, p, x, K, R, D, coef, gout, logits, zq=None, qmu=None, qsg=None, q_ls=False, want_gp=True):
    gp = raw.empty(K, R, D) if want_gp else None
    Dq = zq.shape[-1] if zq is not None else 1
    gqm = raw.empty(R, Dq) if zq is not None else None
    gqs = raw.empty(R, Dq) if zq is not None else None
    g = raw.t(gout)
    raw.call("zs_bernoulli_iw_objective_bwd_f32", raw.t(p), int(logits), raw.t(x), int(x.size), K, R, D, raw.t(coef), g,
             0 if g.numel() == 1 else 1, gp, raw.t(zq), raw.t(qmu), raw.t(qsg), Dq, int(q_ls), gqm, gqs)
    return dict(gp=None if gp is None else gp.cpu().numpy(), gqmu=None if gqm is None else gqm.cpu().numpy(),
                gqsigma=None if gqs is None else gqs.cpu().numpy())


def composed_bwd(raw, p, x, K, R, D, coef, gout, logits, zq=None, qmu=None, qsg=None, q_ls=False):
    dt = np.float32 if raw.dtype == torch.float32 else np.float64
    g = np.asarray(gout, dtype=dt).reshape(-1)
    rowg = (coef.astype(dt) * (g[0] if g.size == 1 else g.reshape(1, R, 1))).astype(dt)        # [2, R, K]
    gp = raw.empty(K, R, D)
    name = "zs_bernoulli_logits_logprob_bwd_f32" if logits else "zs_bernoulli_logprob_bwd_f32"
    raw.call(name, raw.t(p), raw.t(x), int(x.size), raw.t(rowg[0]), 1, K, gp, K, R, D)
    out = dict(gp=gp.cpu().numpy(), gqmu=None, gqsigma=None)
    if zq is not None:
        Dq = zq.shape[-1]
        gm, gs = raw.empty(R, Dq), raw.empty(R, Dq)
        raw.call("zs_normal_logprob_bwd_ksum_f32", raw.t(zq), raw.t(qmu), raw.t(qsg), raw.t(rowg[1]), 1, K, None, gm, gs, K, R, Dq, int(q_ls))
        out.update(gqmu=gm.cpu().numpy(), gqsigma=gs.cpu().numpy())
    return out


def _bwd_inputs(rng, K, R, D, Dq, logits, x_full, q_ls):
    p, x, zq, qmu, qsg, _, _ = _inputs(rng, K, R, D, Dq, logits, x_full, False, False, q_ls)
    coef = rng.standard_normal((2, R, K)) / R
    return p, x, zq, qmu, qsg, coef


def test_c_oracle_iw1_backward_is_the_composition(orc):
    rng = np.random.RandomState(11)
    for K, R, D, Dq, logits, x_full, q_ls in ((5, 3, 256, 8, False, False, False), (4, 2, 64, 3, True, True, True)):
        p, x, zq, qmu, qsg, coef = [a.astype(np.float32) for a in _bwd_inputs(rng, K, R, D, Dq, logits, x_full, q_ls)]
        for gout in (np.float32([0.7]), rng.standard_normal(R).astype(np.float32)):
            got = iw1_bwd(orc, p, x, K, R, D, coef, gout, logits, zq, qmu, qsg, q_ls)
            ref = composed_bwd(orc, p, x, K, R, D, coef, gout, logits, zq, qmu, qsg, q_ls)
            for key in ("gp", "gqmu", "gqsigma"):
                np.testing.assert_array_equal(got[key], ref[key], err_msg=key)
        only_q = iw1_bwd(orc, p, x, K, R, D, coef, np.float32([1.0]), logits, zq, qmu, qsg, q_ls, want_gp=False)
        assert only_q["gp"] is None and np.isfinite(only_q["gqmu"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("K,R,D,Dq,logits,x_full,q_ls", [
    (5, 8, 784, 40, False, False, False), (50, 256, 784, 40, False, False, False), (50, 37, 784, 40, True, False, True),
    (64, 3, 1024, 13, False, True, False), (3, 5, 100, 6, False, False, False), (2, 700, 784, 40, True, False, False),
    (50, 1000, 784, 40, False, False, False)])
def test_hip_iw1_backward(hip, orc, K, R, D, Dq, logits, x_full, q_ls):
    rng = np.random.RandomState(K + R + D)
    p, x, zq, qmu, qsg, coef = [a.astype(np.float32) for a in _bwd_inputs(rng, K, R, D, Dq, logits, x_full, q_ls)]
    for gout in (np.float32([1.0]), np.float32([-0.37]), rng.standard_normal(R).astype(np.float32)):
        got = iw1_bwd(hip, p, x, K, R, D, coef, gout, logits, zq, qmu, qsg, q_ls)
        base = composed_bwd(hip, p, x, K, R, D, coef, gout, logits, zq, qmu, qsg, q_ls)     # same kernels, product formed on the host
        if K * R <= 32768 and D % 4 == 0 and 256 <= D <= 1024:
            np.testing.assert_array_equal(got["gp"], base["gp"])      # the merged launch's Bernoulli role IS K3's wave-per-row backward
        else:
            np.testing.assert_allclose(got["gp"], base["gp"], rtol=1e-6, atol=1e-30)
        for key in ("gqmu", "gqsigma"):                                # (4 instead of 16 K-slices per workgroup: another order of summation)
            np.testing.assert_allclose(got[key], base[key], rtol=2e-5, atol=2e-6 * np.abs(base[key]).max(), err_msg=key)
        only_p = iw1_bwd(hip, p, x, K, R, D, coef, gout, logits)       # the Bernoulli gradient alone takes K3's own launch
        np.testing.assert_array_equal(only_p["gp"], base["gp"])
        if K * R * D <= 2_000_000:
            ref = iw1_bwd(orc, p, x, K, R, D, coef, gout, logits, zq, qmu, qsg, q_ls)
            np.testing.assert_allclose(got["gp"], ref["gp"], rtol=2e-5, atol=1e-6 * np.abs(ref["gp"]).max())
            np.testing.assert_allclose(got["gqmu"], ref["gqmu"], rtol=2e-4, atol=2e-5 * np.abs(ref["gqmu"]).max())
            np.testing.assert_allclose(got["gqsigma"], ref["gqsigma"], rtol=2e-4, atol=2e-5 * np.abs(ref["gqsigma"]).max())


@pytest.mark.gpu
def test_hip_iw1_f64(hip64, orc64):
    rng = np.random.RandomState(5)
    for (K, R, D, Dz, logits, x_full, pms, pss, ls, with_z, with_rows) in CASES[:4]:
        R = min(R, 9)
        p, x, z, pmu, psg, rows_a, logq = _inputs(rng, K, R, D, Dz, logits, x_full, pms, pss, ls)
        args = (p, x, K, R, D, z if with_z else None, pmu if with_z else None, psg if with_z else None, ls, rows_a if with_rows else None, logq)
        for est in (0, 1):
            got, ref = iw1(hip64, *args, est, True, logits), iw1(orc64, *args, est, True, logits)
            np.testing.assert_allclose(got["mean"], ref["mean"], rtol=1e-10)
            np.testing.assert_allclose(got["lp_x"], ref["lp_x"], rtol=1e-10)
            np.testing.assert_allclose(got["coef"], ref["coef"], rtol=1e-6, atol=1e-9)
        coef = rng.standard_normal((2, R, K))
        g1, g2 = iw1_bwd(hip64, p, x, K, R, D, coef, np.float64([0.3]), logits, z, rng.standard_normal((R, Dz)), np.abs(psg) * np.ones((R, Dz)) + 0.5), None
        assert np.isfinite(g1["gp"]).all() and np.isfinite(g1["gqmu"]).all()


# ------------------------------------------------------------------ product level: the fused path == the per-node path
@pytest.mark.parametrize("estimator", ["vimco", "sgvb"])
@pytest.mark.parametrize("fused_logits", [False, True])
def test_fused_generator_side_equals_the_per_node_path(dev, estimator, fused_logits, monkeypatch):
    import zhusuan as zs
    from examples import iwae
    from zhusuan.variational.importance_weighted_objective import ImportanceWeightedObjective as IWO
    import helpers as H
    B, K, hidden = 6, 5, 16
    model = iwae.build(n_samples=K, estimator=estimator, hidden=hidden, device=dev, fused_logits=fused_logits)
    H.load_params_into(model, 77)
    x, e1, e2 = H.iwae_data(B, K)
    xt = torch.tensor(x, device=dev)

    def run(reduce_mean=True):
        for p in model.parameters():
            p.grad = None
        with zs.inject_epsilon([e1, e2]):
            loss = model({"x": xt}, reduce_mean)
        (loss if loss.dim() == 0 else loss.sum()).backward()
        return loss.detach().cpu().numpy(), [p.grad.detach().cpu().numpy().copy() for p in model.parameters()], \
            model.last_iw_bound.detach().cpu().numpy()

    calls = []
    from zhusuan import _hip as hipmod
    klib = hipmod.lib()
    real = klib.call
    monkeypatch.setattr(klib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    fused = run()
    assert "zs_bernoulli_iw_objective_f32" in calls and "zs_bernoulli_logprob_f32" not in calls and "zs_iw_objective_f32" not in calls
    fused_vec = run(False) if estimator == "sgvb" else None
    monkeypatch.setattr(IWO, "_generator_side_in_one_launch", lambda self, *a: None)
    calls.clear()
    plain = run()
    assert "zs_bernoulli_iw_objective_f32" not in calls and "zs_iw_objective_f32" in calls
    np.testing.assert_allclose(fused[0], plain[0], rtol=2e-6)
    np.testing.assert_allclose(fused[2], plain[2], rtol=2e-6)
    for a, b in zip(fused[1], plain[1]):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * max(np.abs(b).max(), 1e-30))   # (VIMCO's signal amplifies the rows' last-bit differences)
    if fused_vec is not None:
        plain_vec = run(False)
        assert fused_vec[0].shape == plain_vec[0].shape == (B,)
        np.testing.assert_allclose(fused_vec[0], plain_vec[0], rtol=2e-6)
        for a, b in zip(fused_vec[1], plain_vec[1]):
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * max(np.abs(b).max(), 1e-30))   # (VIMCO's signal amplifies the rows' last-bit differences)


def test_staged_backward_launches_only_the_side_each_stage_needs(dev, monkeypatch):
    """dataparallel.StagedBuckets runs backward once per stage (decoder parameters, then encoder parameters) on a retained
    graph.  IW1's backward node lies on both paths; told the targets of the pass it launches the Bernoulli side in stage 0 and
    the log q side in stage 1 -- not both in both -- and the gradients equal the single-pass ones."""
    import zhusuan as zs
    from zhusuan import dataparallel, _hip as hipmod
    from examples import iwae
    import helpers as H
    B, K = 6, 5
    model = iwae.build(n_samples=K, estimator="vimco", hidden=16, device=dev)
    H.load_params_into(model, 31)
    x, e1, e2 = H.iwae_data(B, K)
    xt = torch.tensor(x, device=dev)
    with zs.inject_epsilon([e1, e2]):
        loss = model({"x": xt})
    loss.backward()
    ref = [p.grad.detach().clone() for p in model.parameters()]
    sb = dataparallel.StagedBuckets([model.generator.parameters(), model.variational.parameters()])
    sb.zero()
    with zs.inject_epsilon([e1, e2]):
        loss = model({"x": xt})
    klib = hipmod.lib()
    real, calls = klib.call, []

    def spy(name, *a):
        if name.startswith("zs_bernoulli_iw_objective_bwd"):
            calls.append((a[10] is not None, a[11] is not None))          # (gp wanted, q side wanted)
        return real(name, *a)
    monkeypatch.setattr(klib, "call", spy)
    sb.backward_stage(loss, 0)
    assert calls == [(True, False)]
    sb.backward_stage(loss, 1)
    assert calls == [(True, False), (False, True)]
    for p, r in zip(model.parameters(), ref):
        # (the merged single-pass launch sums the K particles in 4 slices, the stand-alone log q launch in 16: last-bit differences)
        np.testing.assert_allclose(p.grad.detach().cpu().numpy(), r.cpu().numpy(), rtol=2e-4, atol=1e-6 * float(r.abs().max()))


@pytest.mark.gpu
def test_hip_iw1_launches_on_two_streams_at_once(hip):
    """The batch mean is finished by a wave of workgroup 0 that WATCHES the shard words (zs_iwpersist.h: iw1_watch) while the other
    workgroups of its launch may not even be resident yet.  Two launches in flight at once -- two streams, an accumulator each,
    256 + 256 workgroups of 1 024 threads competing for the same CUs, 40 rounds without a synchronisation in between -- must each
    end with their own mean, the one a launch alone on the device gives, and with their accumulators back at zero."""
    K, D = 50, 784
    shapes = (300, 700)                     # more datapoints than CUs on both: every CU is wanted by both launches
    rng = np.random.RandomState(11)
    P = _hip.ptr
    sets = []
    for R in shapes:
        p, x, z, pmu, psg, rows_a, logq = [a.astype(np.float32) for a in _inputs(rng, K, R, D, 40, False, False, False, False, False)]
        alone = iw1(hip, p, x, K, R, D, z, pmu, psg, False, None, logq, 1, True, False)
        t = dict(p=hip.t(p), x=hip.t(x), z=hip.t(z), pmu=hip.t(pmu), psg=hip.t(psg), logq=hip.t(logq),
                 lp_x=hip.empty(R, K), lp_z=hip.empty(R, K), cost_b=hip.empty(R), bound=hip.empty(R), coef=hip.empty(2, R, K),
                 means=hip.empty(40), acc=torch.zeros(64, dtype=torch.int64, device=hip.dev), R=R, stream=torch.cuda.Stream(hip.dev))
        sets.append((t, alone))
    torch.cuda.synchronize()
    for i in range(40):
        for t, _ in sets:
            R = t["R"]
            hip.k.call("zs_bernoulli_iw_objective_f32", P(t["p"]), 0, P(t["x"]), R * D, K, R, D, P(t["z"]), P(t["pmu"]), R * 40, P(t["psg"]), R * 40,
                       40, 0, None, K, P(t["logq"]), K, 1, 1, P(t["lp_x"]), P(t["lp_z"]), P(t["cost_b"]), P(t["bound"]), P(t["coef"]),
                       P(t["means"][i:]), P(t["acc"]), ctypes.c_void_p(t["stream"].cuda_stream))
    torch.cuda.synchronize()
    for t, alone in sets:
        means = t["means"].cpu().numpy()
        assert (means == alone["mean"][0]).all(), (means, alone["mean"][0])
        assert int(t["acc"].abs().sum().item()) == 0
        np.testing.assert_array_equal(t["cost_b"].cpu().numpy(), alone["cost"])


def test_accumulator_health_helpers(dev):
    """A watcher that gives up (it cannot, in a healthy process) stores NaN, leaves the words alone and raises the accumulator's
    poison word (include/zs_hip.h); `zhusuan._ops.iw1_accumulators_ok()` / `reset_iw1_accumulators()` are the owner's way to
    see it and to re-zero.  Here: healthy after an evaluation; a raised word is seen; reset clears every word."""
    import zhusuan as zs
    from zhusuan import _ops
    from examples import iwae
    model = iwae.build(n_samples=5, estimator="vimco", hidden=16, device=dev)
    model({"x": (torch.rand(4, 784, device=dev) < 0.5).float()})
    assert zs.explain(model).startswith("IW1") and _ops.iw1_accumulators_ok()
    accs = [sc[0] for key, sc in _ops._SCRATCH.items() if key[2] == "iw1"]
    assert accs and all(int(a.abs().sum()) == 0 for a in accs)          # handed back at zero
    accs[0][_ops.IW1_POISON_WORD] = 1
    accs[0][3] = 12345
    assert not _ops.iw1_accumulators_ok()
    _ops.reset_iw1_accumulators()
    assert _ops.iw1_accumulators_ok() and all(int(a.abs().sum()) == 0 for a in accs)


@pytest.mark.gpu
def test_lab_build_applies_and_its_last_arrival_finish_returns_the_watchers_bits(hip):
    """The kernel laboratory is a patch on the release sources (tools/lab/csrc_lab.patch, `make experiments`): it must keep applying
    and building.  And the two finishes of IW1's batch mean -- the release build's watching wave, the lab build's last arrival
    (ZS_IW1_SHARDED=1: a returning atomic per level) -- add the same fixed-point shares: every output, the mean included, must be
    bit-identical (an independent check of the watcher, which the strict-ordering twin does not cover)."""
    import glob
    import os
    import subprocess
    from conftest import ROOT
    csrc = os.path.join(ROOT, "zhusuan-pytorch_amd", "csrc")
    exp = os.path.join(ROOT, "tools", "_exp", "libzs_hip_exp.so")
    srcs = glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + \
        [os.path.join(ROOT, "include", "zs_hip.h"), os.path.join(ROOT, "tools", "lab", "csrc_lab.patch")]
    if not os.path.exists(exp) or os.path.getmtime(exp) < max(os.path.getmtime(f) for f in srcs):
        r = subprocess.run(["make", "-C", csrc, "experiments"], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, "make experiments failed (does tools/lab/csrc_lab.patch still apply?):\n" + (r.stdout + r.stderr)[-3000:]
    os.environ["ZS_IW1_SHARDED"] = "1"          # read once, at the lab library's first IW1 launch; the release library reads nothing
    try:
        lab = Raw(_hip.KernelLibrary(exp), "cuda:0")
        assert "experiments" in lab.k.build_info() and "experiments" not in hip.k.build_info()
        rng = np.random.RandomState(4)
        first = True
        for (K, R, D, Dz) in [(50, 256, 784, 40), (50, 300, 784, 40), (5, 8, 784, 40), (17, 1000, 512, 8), (64, 37, 1024, 12), (2, 4096, 256, 4)]:
            p, x, z, pmu, psg, rows_a, logq = _inputs(rng, K, R, D, Dz, False, False, False, False, False)
            for est in (0, 1):
                a = iw1(hip, p, x, K, R, D, z, pmu, psg, False, None, logq, est, True, False)
                b = iw1(lab, p, x, K, R, D, z, pmu, psg, False, None, logq, est, True, False)
                if first:
                    os.environ.pop("ZS_IW1_SHARDED", None)
                    first = False
                for key in ("lp_x", "lp_z", "cost", "bound", "coef", "mean"):
                    assert np.array_equal(a[key], b[key]), (K, R, D, est, key)
                assert np.isfinite(a["mean"]).all()
    finally:
        os.environ.pop("ZS_IW1_SHARDED", None)
