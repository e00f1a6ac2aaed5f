"""Pins oracle/zs_oracle.py (the torch-CPU restatement) to the reference's own outputs.

The fixtures were produced by tests/golden/gen_golden.py from the real reference.
CPU only; runs in the default (-m "not gpu") suite.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers as H
from oracle import zs_oracle as O

torch.set_num_threads(8)


def T(a, rg=False):
    x = torch.tensor(np.asarray(a, dtype=np.float32))
    return x.requires_grad_(rg)


def close(a, b, rtol=2e-6, atol=2e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_normal_sample_logprob_cases():
    g = load_golden("g_normal_sample")
    n = int(g["n_cases"])
    assert n == 72
    for c in range(n):
        p = "c%03d_" % c
        K = int(g[p + "K"])
        K = None if K < 0 else K
        reparam = bool(g[p + "reparam"])
        gnd = int(g[p + "g"])
        mu, ls = T(g[p + "mu"], True), T(g[p + "ls"], True)
        sd = torch.exp(ls)
        np.testing.assert_allclose(sd.detach().numpy(), g[p + "sd"], rtol=2e-7, atol=0)
        with torch.no_grad():
            sd_exact = T(g[p + "sd"])      # forward with the stored std (bit-exact z), gradients through exp(ls)
        sd = sd + (sd_exact - sd).detach()
        z = O.normal_sample(mu, sd, T(g[p + "eps"]), K, reparam)
        lp = O.normal_log_prob(mu, sd, z, gnd)
        assert tuple(z.shape) == g[p + "z"].shape
        assert np.array_equal(z.detach().numpy(), g[p + "z"]), "sample must be bit-exact (case %d)" % c
        close(lp, g[p + "lp"])
        obj = (lp * T(g[p + "w"])).sum() + (z * T(g[p + "wz"])).sum()
        gmu, gls = torch.autograd.grad(obj, [mu, ls], allow_unused=True)
        close(gmu, g[p + "gmu"], 2e-5, 2e-5)
        close(gls, g[p + "gls"], 2e-5, 2e-5)


def test_normal_logprob_given_value():
    g = load_golden("g_normal_logprob")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        mu, sd, x = T(g[p + "mu"], True), T(g[p + "sd"], True), T(g[p + "x"], True)
        lp = O.normal_log_prob(mu, sd, x, int(g[p + "g"]))
        close(lp, g[p + "lp"])
        gmu, gsd, gx = torch.autograd.grad((lp * T(g[p + "w"])).sum(), [mu, sd, x])
        close(gmu, g[p + "gmu"], 1e-5, 1e-5)
        close(gsd, g[p + "gsd"], 1e-5, 1e-5)
        close(gx, g[p + "gx"], 1e-5, 1e-5)


def test_normal_eps_has_mean_shape():
    g = load_golden("g_normal_epsshape")
    assert tuple(g["draw_shape"]) == (2, 1, 3)
    z = O.normal_sample(T(g["mu"]), T(g["sd"]), T(g["eps"]), 2)
    assert np.array_equal(z.numpy(), g["z"])
    close(O.normal_log_prob(T(g["mu"]), T(g["sd"]), z), g["lp"])


def test_normal_known_answers():
    # reference test/distributions/test_normal.py:92-126 (scipy logpdf, rtol 1e-3) and SURVEY 7.4-4
    lp = O.normal_log_prob(T([1.]), torch.exp(T([-3., -1., 1., 10.])), T([0.99, 0.9, 9., 99.]))
    close(lp, [2.06088996, 0.04411618, -6.24966812, -10.91894817], 1e-6, 1e-6)
    # docs/tutorials/concepts.rst:70-73
    lp = O.normal_log_prob(T([[-1., 1.], [0., -2.]]), T(1.), torch.zeros([1]), 1)
    close(lp, [-2.83787704, -3.83787727], 1e-6, 1e-6)


def test_bernoulli_cases():
    g = load_golden("g_bernoulli")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        gnd = int(g[p + "g"])
        if int(g[p + "from_logits"]):
            lg = T(g[p + "logits"], True)
            pr = O.bernoulli_probs_from_logits(lg)
            close(pr, g[p + "probs"], 1e-6, 1e-7)
            lp = O.bernoulli_log_prob(pr, T(g[p + "x"]), gnd)
            close(lp, g[p + "lp"])
            (gl,) = torch.autograd.grad((lp * T(g[p + "w"])).sum(), [lg])
            close(gl, g[p + "gl"], 1e-5, 1e-5)
        else:
            pr = T(g[p + "probs"], True)
            lp = O.bernoulli_log_prob(pr, T(g[p + "x"]), gnd)
            close(lp, g[p + "lp"], 2e-6, 1e-4 if gnd else 2e-6)
            if (p + "gp") in g.files:
                (gp,) = torch.autograd.grad((lp * T(g[p + "w"])).sum(), [pr])
                close(gp, g[p + "gp"], 1e-5, 1e-5)
    close(O.bernoulli_logits_from_probs(T(g["ctor_probs"])), g["ctor_logits"], 1e-6, 1e-7)
    assert float(g["ctor_zero_logit_probs"]) == 0.5
    # edge values quoted in SURVEY 7.4-4: p in {0, 1} gives 0 / -18.420681
    lp = O.bernoulli_log_prob(T([0., 1.]), T([1., 1.]))
    close(lp, [-18.420681, 0.0], 1e-6, 1e-6)
    lp = O.bernoulli_log_prob(O.bernoulli_probs_from_logits(T([2., 1.])), T([0., 1.]))
    close(lp, [-2.12692761, -0.31326166], 1e-6, 1e-6)


def test_stochastic_tensor_reductions():
    g = load_golden("g_stochastic_tensor")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        K = int(g[p + "K"])
        K = None if K < 0 else K
        mu, sd = T(g[p + "mu"]), T(g[p + "sd"])
        first = O.normal_sample(mu, sd, T(g[p + "e1"]), K)
        second = O.normal_sample(mu, sd, T(g[p + "e2"]), K)
        assert np.array_equal(first.numpy(), g[p + "first"])
        assert np.array_equal(second.numpy(), g[p + "second"])
        rm = [int(v) for v in g[p + "rm"]] or None
        rs = [int(v) for v in g[p + "rs"]] or None
        mult = float(g[p + "mult"]) or None
        lp = O.st_reduce(O.normal_log_prob(mu, sd, second, int(g[p + "g"])), rm, rs, mult)
        assert tuple(lp.shape) == tuple(int(v) for v in g[p + "lp_shape"])
        close(lp, g[p + "lp"], 3e-6, 3e-6)


def test_log_mean_exp():
    g = load_golden("g_log_mean_exp")
    close(O.log_mean_exp(T(g["a_x"]), 0), g["a_dim0"])
    close(O.log_mean_exp(T(g["a_x"]), 0), [2.43378091, 1.30685282], 1e-6, 1e-6)
    close(O.log_mean_exp(T(g["a_x"]), 1, True), g["a_dim1_keep"])
    close(O.log_mean_exp(T(g["b_x"]), 0), g["b_dim0"])
    close(O.log_mean_exp(T(g["b_x"]), 1), g["b_dim1"])
    close(O.log_mean_exp(T(g["b_x"]), 2, True), g["b_dim2_keep"])


def test_iw_estimators():
    g = load_golden("g_iw")
    for c in range(int(g["n_cases"])):
        p = "c%03d_" % c
        for est, fn in [("sgvb", O.iw_sgvb), ("vimco", O.iw_vimco)]:
            a, b = T(g[p + "logp"], True), T(g[p + "logq"], True)
            cost = fn(a, b, 0)
            ga, gb = torch.autograd.grad(cost, [a, b])
            close(cost, g[p + est + "_cost"], 1e-6, 1e-6)
            close(ga, g[p + est + "_glogp"], 1e-5, 1e-7)
            # d/dlogq of vimco carries the fp32 cancellation noise of the reference itself
            # (SURVEY 7.4-6): compare within the reference's own distance to its float64 run.
            ref_err = np.abs(g[p + est + "_glogq"] - g[p + est + "_glogq64"]).max()
            np.testing.assert_allclose(gb.numpy(), g[p + est + "_glogq"], rtol=1e-5, atol=max(4 * ref_err, 1e-7))
        close(O.iw_sgvb(T(g[p + "logp"]), T(g[p + "logq"]), 0, False), g[p + "sgvb_cost_noreduce"], 1e-6, 1e-6)
        close(O.log_mean_exp(T(g[p + "logp"]) - T(g[p + "logq"]), 0), g[p + "bound"], 1e-6, 1e-6)
    for est, fn in [("sgvb", O.iw_sgvb), ("vimco", O.iw_vimco)]:
        a, b = T(g["d1_logp"], True), T(g["d1_logq"], True)
        cost = fn(a, b, 0)
        ga, gb = torch.autograd.grad(cost, [a, b])
        close(cost, g["d1_" + est + "_cost"], 1e-6, 1e-6)
        close(ga, g["d1_" + est + "_glogp"], 1e-5, 1e-7)
        close(gb, g["d1_" + est + "_glogq"], 1e-4, 1e-6)


def test_iw_c3_scalars():
    g = load_golden("g_iw_c3")
    for i in range(3):
        r2 = np.random.RandomState(4100 + i)
        spread = float(g["s%d_spread" % i])
        logp = (-550.0 + spread * r2.standard_normal((50, 256))).astype(np.float32)
        logq = (-50.0 + 0.3 * spread * r2.standard_normal((50, 256))).astype(np.float32)
        for est, fn in [("sgvb", O.iw_sgvb), ("vimco", O.iw_vimco)]:
            a, b = T(logp, True), T(logq, True)
            c = fn(a, b, 0)
            ga, gb = torch.autograd.grad(c, [a, b])
            close(c, g["s%d_%s_cost" % (i, est)], 1e-6, 0)
            close(ga.sum(), g["s%d_%s_glogp_sum" % (i, est)], 1e-5, 1e-6)
            close(gb.abs().sum(), g["s%d_%s_glogq_abs_sum" % (i, est)], 1e-4, 1e-6)
        close(O.log_mean_exp(T(logp) - T(logq), 0).mean(), g["s%d_bound_mean" % i], 1e-6, 0)


def test_elbo_sgvb():
    g = load_golden("g_elbo_sgvb")
    a, b = T(g["logp"]), T(g["logq"])
    close(O.elbo_sgvb(a, b, True), g["sgvb_mean"], 1e-6, 0)
    close(O.elbo_sgvb(a, b, False), g["sgvb_nomean"], 1e-6, 0)
    close(O.elbo_sgvb(a[0, 0], b[0, 0], True), g["sgvb_scalar"], 1e-6, 0)


def _check_grads(g, named):
    names = [str(n) for n in g["grad_names"]]
    assert names == [n for n, _ in named]
    norms, sums = H.grad_stats(named)
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-6)
    for (n, gr) in named:
        np.testing.assert_allclose(gr.reshape(-1)[:8].numpy(), g["ghead_" + n], rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("tag,B", [("small", 8), ("c1", 64), ("c2", 512)])
def test_vae_end_to_end(tag, B):
    g = load_golden("g_vae_" + tag)
    spec = H.vae_param_spec()
    p = H.make_params(spec, 1000 + B)
    x, e1, e2 = H.vae_data(B)
    assert [tuple(d) for d in g["draws"]] == [(B, 40), (B, 40)]  # two draws, second one used
    loss, aux = O.vae_loss(p, T(x), T(e2))
    close(loss, g["loss"], 2e-6, 0)
    close(aux["logqz"], g["logqz"], 1e-5, 0)
    close(aux["logpz"], g["logpz"], 1e-5, 0)
    close(aux["logpx"], g["logpx"], 1e-5, 0)
    grads = torch.autograd.grad(loss, [p[n] for n, _ in spec])
    _check_grads(g, [(n, gr) for (n, _), gr in zip(spec, grads)])
    if tag == "small":
        assert np.array_equal(aux["z"].detach().numpy(), g["z"])
        close(aux["x_mean"], g["x_mean"], 1e-5, 1e-6)


@pytest.mark.parametrize("est", ["sgvb", "vimco"])
@pytest.mark.parametrize("tag,B,K,hidden", [("small", 8, 5, 32), ("c3", 256, 50, 500)])
def test_iwae_end_to_end(est, tag, B, K, hidden):
    g = load_golden("g_iwae_%s_%s" % (est, tag))
    spec = H.iwae_param_spec(hidden=hidden)
    p = H.make_params(spec, 2000 + B + K)
    x, e1, e2 = H.iwae_data(B, K)
    assert [tuple(d) for d in g["draws"]] == [(K, B, 40), (K, B, 40)]
    loss, aux = O.iwae_loss(p, T(x), T(e2), K, est)
    close(loss, g["loss"], 3e-6, 0)
    close(aux["iw_bound"], g["iw_bound"], 3e-6, 0)
    grads = torch.autograd.grad(loss, [p[n] for n, _ in spec])
    _check_grads(g, [(n, gr) for (n, _), gr in zip(spec, grads)])
    if tag == "small":
        assert np.array_equal(aux["z"].detach().numpy(), g["z"])
    close(aux["log_w"], g["log_w"], 1e-5, 1e-4)          # every log-importance-weight, small and config shape


@pytest.mark.parametrize("tag,B,K", [("small", 16, 4), ("c5", 512, 10), ("c5g", 4096, 10)])
def test_bnn_end_to_end(tag, B, K):
    g = load_golden("g_bnn_" + tag)
    assert int(g["n_draws"]) == 4 and tuple(g["draws_w0"]) == (K, 50, 14) and tuple(g["draws_w1"]) == (K, 1, 51)
    x, y, eps = H.bnn_data(B, K)
    wm, wl, yl = H.bnn_params(B, K)
    loss, aux = O.bnn_loss(wm, wl, yl, T(x), T(y), [T(eps[2]), T(eps[3])], K)
    close(loss, g["loss"], 3e-6, 0)
    close(aux["rmse"], g["rmse"], 1e-5, 0)
    close(aux["logp_y"], g["logp_y"], 1e-5, 0)
    grads = torch.autograd.grad(loss, wm + wl + [yl])
    for i in range(2):
        close(grads[i], g["g_w_mean_%d" % i], 2e-4, 2e-5)
        close(grads[2 + i], g["g_w_logstd_%d" % i], 2e-4, 2e-5)
    close(grads[4], g["g_y_logstd"], 2e-4, 1e-5)
