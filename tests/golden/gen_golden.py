#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Run in the build container only (the reference lives at /root/reference and never
travels to the GPU box):

    python tests/golden/gen_golden.py

Every fixture is plain data (inputs + the reference's outputs) stored as .npz.
Inputs come from ``np.random.RandomState(seed)``; the reference's Gaussian draws
are replaced by explicit epsilon tensors through a temporary patch of
``torch.normal`` (the reference draws with ``torch.normal(0., 1., size=...)`` in
zhusuan/distributions/normal.py:104 and ``torch.normal(mean, std)`` at :102;
both equal ``mean + std * randn`` bit-for-bit on CPU).

Nothing from the reference is copied here: the script only *calls* it.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.dont_write_bytecode = True   # the reference tree is read-only: no __pycache__ next to its sources
OUT = os.path.dirname(os.path.abspath(__file__))
OUT_SRC = OUT             # (OUT is redirected by tests/test_golden_regeneration.py; the tree's location is not)
sys.path.insert(0, REF)

import zhusuan  # noqa: E402  (the reference)
from zhusuan.distributions import Normal, Bernoulli  # noqa: E402
from zhusuan.framework.bn import BayesianNet  # noqa: E402
from zhusuan.variational.elbo import ELBO  # noqa: E402
from zhusuan.variational.importance_weighted_objective import (  # noqa: E402
    ImportanceWeightedObjective,
)

assert zhusuan.__file__.startswith(REF), zhusuan.__file__
torch.set_num_threads(8)
F32 = np.float32


# --------------------------------------------------------------------------
# epsilon injection
# --------------------------------------------------------------------------
class EpsQueue(object):
    """Replaces torch.normal while the reference runs; records every draw."""

    def __init__(self, eps_list):
        self.eps = [torch.as_tensor(e) for e in eps_list]
        self.calls = []
        self._orig = None

    def _normal(self, mean, std=None, size=None, **kw):
        e = self.eps.pop(0)
        if size is not None:  # torch.normal(0., 1., size=shape)
            assert tuple(size) == tuple(e.shape), (tuple(size), tuple(e.shape))
            self.calls.append(("std", tuple(size)))
            return e.clone()
        # torch.normal(mean_tensor, std_tensor): detached draw
        shp = torch.broadcast_shapes(mean.shape, std.shape)
        assert tuple(shp) == tuple(e.shape), (tuple(shp), tuple(e.shape))
        self.calls.append(("loc", tuple(shp)))
        return (mean + std * e).detach()

    def __enter__(self):
        self._orig = torch.normal
        torch.normal = self._normal
        return self

    def __exit__(self, *a):
        torch.normal = self._orig
        assert not self.eps, "unused epsilon tensors: %d" % len(self.eps)


def t(a, requires_grad=False):
    x = torch.tensor(np.asarray(a, dtype=F32))
    x.requires_grad_(requires_grad)
    return x


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    clean = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        clean[k] = np.asarray(v)
    np.savez_compressed(path, **clean)
    print("wrote %-28s %7.1f KB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024.0, len(clean)))


# --------------------------------------------------------------------------
# G-N1: Normal sample + log_prob
# --------------------------------------------------------------------------
def gen_normal():
    rng = np.random.RandomState(101)
    out = {}
    case = 0
    for shape in [(8, 4), (3, 5, 8)]:
        for K in [None, 1, 5]:
            for reparam in [True, False]:
                for use_logstd in [False, True]:
                    for g in [0, 1, 2]:
                        mu = rng.standard_normal(shape).astype(F32)
                        ls = (0.3 * rng.standard_normal(shape) - 0.5).astype(F32)
                        eshape = shape if (K is None or K == 1) else (K,) + shape
                        eps = rng.standard_normal(eshape).astype(F32)
                        mu_t, ls_t = t(mu, True), t(ls, True)
                        if use_logstd:
                            d = Normal(mean=mu_t, logstd=ls_t, is_reparameterized=reparam, group_ndims=g)
                        else:
                            d = Normal(mean=mu_t, std=torch.exp(ls_t), is_reparameterized=reparam, group_ndims=g)
                        with EpsQueue([eps]):
                            z = d.sample(K)
                        lp = d.log_prob(None)  # uses sample_cache (normal.py:110)
                        w = rng.standard_normal(tuple(lp.shape)).astype(F32)
                        wz = rng.standard_normal(tuple(z.shape)).astype(F32)
                        obj = (lp * t(w)).sum() + (z * t(wz)).sum()
                        gmu, gls = torch.autograd.grad(obj, [mu_t, ls_t], allow_unused=True)
                        p = "c%03d_" % case
                        out[p + "mu"], out[p + "ls"], out[p + "eps"] = mu, ls, eps
                        out[p + "sd"] = d.std   # exp(ls) as evaluated HERE (CPU exp differs by an ulp across ISAs)
                        out[p + "K"] = np.array(-1 if K is None else K)
                        out[p + "reparam"] = np.array(int(reparam))
                        out[p + "use_logstd"] = np.array(int(use_logstd))
                        out[p + "g"] = np.array(g)
                        out[p + "z"], out[p + "lp"] = z, lp
                        out[p + "w"], out[p + "wz"] = w, wz
                        out[p + "gmu"] = gmu if gmu is not None else np.zeros(shape, F32)
                        out[p + "gls"] = gls if gls is not None else np.zeros(shape, F32)
                        case += 1
    out["n_cases"] = np.array(case)
    save("g_normal_sample", **out)

    # log_prob of a *given* value with leading-axis / scalar broadcast (prior & likelihood uses)
    out = {}
    case = 0
    specs = [
        # (mean shape, std shape, given shape, group_ndims)
        ((6, 4), (6, 4), (6, 4), 0),
        ((6, 4), (6, 4), (3, 6, 4), 0),     # IWAE prior: params broadcast over K (normal.py:112-116)
        ((6, 4), (6, 4), (3, 6, 4), 1),
        ((5, 7), (5, 7), (4, 5, 7), 2),     # BNN prior on w with group_ndims=2 (bnn_vi.py:32)
        ((4, 6), (1,), (6,), 0),            # BNN likelihood: mean [K,B], std [1], y [B] (bnn_vi.py:55)
        ((1, 3), (2, 1), (2, 3), 0),        # middle broadcast
        ((2,), (2, 2), (1,), 0),            # test_normal.py:126 style
    ]
    for ms, ss, gs, g in specs:
        mu = rng.standard_normal(ms).astype(F32)
        sd = np.exp(0.4 * rng.standard_normal(ss)).astype(F32)
        x = (2.0 * rng.standard_normal(gs)).astype(F32)
        mu_t, sd_t, x_t = t(mu, True), t(sd, True), t(x, True)
        d = Normal(mean=mu_t, std=sd_t, group_ndims=g)
        lp = d.log_prob(x_t)
        w = rng.standard_normal(tuple(lp.shape)).astype(F32)
        gmu, gsd, gx = torch.autograd.grad((lp * t(w)).sum(), [mu_t, sd_t, x_t])
        p = "c%03d_" % case
        out[p + "mu"], out[p + "sd"], out[p + "x"], out[p + "g"] = mu, sd, x, np.array(g)
        out[p + "lp"], out[p + "w"] = lp, w
        out[p + "gmu"], out[p + "gsd"], out[p + "gx"] = gmu, gsd, gx
        case += 1
    out["n_cases"] = np.array(case)
    save("g_normal_logprob", **out)

    # epsilon-sharing quirk: eps has mean's shape (normal.py:91-92,104)
    mu = rng.standard_normal((1, 3)).astype(F32)
    sd = np.exp(rng.standard_normal((2, 1))).astype(F32)
    eps = rng.standard_normal((2, 1, 3)).astype(F32)
    d = Normal(mean=t(mu), std=t(sd))
    with EpsQueue([eps]) as q:
        z = d.sample(2)
    save("g_normal_epsshape", mu=mu, sd=sd, eps=eps, z=z, lp=d.log_prob(None),
         draw_shape=np.array(q.calls[0][1]))


# --------------------------------------------------------------------------
# G-B1: Bernoulli log_prob
# --------------------------------------------------------------------------
def gen_bernoulli():
    rng = np.random.RandomState(202)
    out = {}
    case = 0
    edge = np.array([0.0, 1e-9, 1e-8, 1e-4, 0.2, 0.25, 0.5, 0.75, 0.8, 1 - 1e-4, 1 - 1e-7, 1.0], F32)
    # edge probabilities against x in {0,1} and fractional x
    for xs in [np.zeros_like(edge), np.ones_like(edge), np.full_like(edge, 0.3)]:
        d = Bernoulli(probs=t(edge))
        lp = d.log_prob(t(xs))
        p = "c%03d_" % case
        out[p + "probs"], out[p + "x"], out[p + "lp"] = edge, xs, lp
        out[p + "from_logits"] = np.array(0)
        out[p + "g"] = np.array(0)
        case += 1
    # random, K-broadcast (x [B,X] against p [K,B,X], bernoulli.py:88-92), grads wrt probs
    for pshape, xshape, g in [((6, 16), (6, 16), 0), ((3, 6, 16), (6, 16), 0), ((3, 6, 16), (6, 16), 1),
                              ((2, 5, 784), (5, 784), 0)]:
        pr = rng.uniform(0.001, 0.999, pshape).astype(F32)
        x = (rng.uniform(size=xshape) < 0.5).astype(F32)
        pr_t = t(pr, True)
        d = Bernoulli(probs=pr_t, group_ndims=g)
        lp = d.log_prob(t(x))
        w = rng.standard_normal(tuple(lp.shape)).astype(F32)
        (gp,) = torch.autograd.grad((lp * t(w)).sum(), [pr_t])
        p = "c%03d_" % case
        out[p + "probs"], out[p + "x"], out[p + "lp"] = pr, x, lp
        out[p + "w"], out[p + "gp"] = w, gp
        out[p + "from_logits"] = np.array(0)
        out[p + "g"] = np.array(g)
        case += 1
    # logits constructor (bernoulli.py:46-50): probs = sigmoid(logits)
    for lshape, xshape in [((2,), (2,)), ((4, 9), (4, 9)), ((3, 4, 9), (4, 9))]:
        lg = (3.0 * rng.standard_normal(lshape)).astype(F32)
        x = (rng.uniform(size=xshape) < 0.5).astype(F32)
        lg_t = t(lg, True)
        d = Bernoulli(logits=lg_t)
        lp = d.log_prob(t(x))
        w = rng.standard_normal(tuple(lp.shape)).astype(F32)
        (gl,) = torch.autograd.grad((lp * t(w)).sum(), [lg_t])
        p = "c%03d_" % case
        out[p + "logits"], out[p + "x"], out[p + "lp"], out[p + "probs"] = lg, x, lp, d.probs
        out[p + "w"], out[p + "gl"] = w, gl
        out[p + "from_logits"] = np.array(1)
        out[p + "g"] = np.array(0)
        case += 1
    out["n_cases"] = np.array(case)
    # constructor identities (test_bernoulli.py:20-27)
    b = Bernoulli(probs=[0.4, 0.5])
    out["ctor_probs"] = b.probs
    out["ctor_logits"] = b.logits
    out["ctor_zero_logit_probs"] = Bernoulli(0.).probs
    save("g_bernoulli", **out)


# --------------------------------------------------------------------------
# G-L1 / G-U1: Logistic and Uniform
# --------------------------------------------------------------------------
class UniformQueue(object):
    """Replaces the U(0,1) draws while the reference runs: ``torch.nn.init.uniform_`` (logistic.py:64) and
    ``torch.rand`` (behind torch.distributions.Uniform.sample, uniform.py:64,66-67)."""

    def __init__(self, u_list):
        self.u = [torch.as_tensor(a) for a in u_list]
        self.calls = []

    def _uniform_(self, tensor, a=0., b=1.):
        assert (a, b) == (0., 1.)
        u = self.u.pop(0)
        assert tuple(tensor.shape) == tuple(u.shape), (tuple(tensor.shape), tuple(u.shape))
        self.calls.append(tuple(u.shape))
        return u.clone()

    def _rand(self, *size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        u = self.u.pop(0)
        assert shape == tuple(u.shape), (shape, tuple(u.shape))
        self.calls.append(shape)
        return u.clone()

    def __enter__(self):
        self._o1, self._o2 = torch.nn.init.uniform_, torch.rand
        torch.nn.init.uniform_ = self._uniform_
        torch.rand = self._rand
        return self

    def __exit__(self, *a):
        torch.nn.init.uniform_, torch.rand = self._o1, self._o2
        assert not self.u, "unused draws: %d" % len(self.u)


def gen_logistic():
    from zhusuan.distributions import Logistic
    rng = np.random.RandomState(707)
    out = {}
    case = 0
    for shape in [(8, 4), (3, 5, 8)]:
        for K in [None, 1, 5]:
            for g in [0, 1, 2]:
                loc = rng.standard_normal(shape).astype(F32)
                sc = np.exp(0.4 * rng.standard_normal(shape) - 0.3).astype(F32)
                ushape = shape if (K is None or K == 1) else (K,) + shape
                u = rng.uniform(1e-4, 1 - 1e-4, ushape).astype(F32)
                loc_t, sc_t = t(loc, True), t(sc, True)
                d = Logistic(loc_t, sc_t, group_ndims=g)
                with UniformQueue([u]):
                    z = d.sample(K)
                lp = d.log_prob(None)   # sample_cache, logistic.py:70-71
                w = rng.standard_normal(tuple(lp.shape)).astype(F32)
                wz = rng.standard_normal(tuple(z.shape)).astype(F32)
                gl, gs = torch.autograd.grad((lp * t(w)).sum() + (z * t(wz)).sum(), [loc_t, sc_t])
                p = "c%03d_" % case
                out[p + "loc"], out[p + "scale"], out[p + "u"] = loc, sc, u
                out[p + "K"] = np.array(-1 if K is None else K)
                out[p + "g"] = np.array(g)
                out[p + "z"], out[p + "lp"], out[p + "w"], out[p + "wz"] = z, lp, w, wz
                out[p + "gloc"], out[p + "gscale"] = gl, gs
                case += 1
    out["n_cases"] = np.array(case)
    save("g_logistic_sample", **out)

    out = {}
    case = 0
    specs = [  # (loc shape, scale shape, given shape, group_ndims)
        ((6, 4), (6, 4), (6, 4), 0),
        ((6, 4), (6, 4), (3, 6, 4), 0),
        ((6, 4), (6, 4), (3, 6, 4), 1),
        ((5, 7), (5, 7), (4, 5, 7), 2),
        ((1, 3), (2, 1), (2, 3), 0),
        ((16,), (16,), (7, 16), 1),          # nice_mnist.py:29 style prior: loc/scale [D], data [B, D]
        ((2,), (2, 2), (1,), 0),
    ]
    for ls, ss, gs, g in specs:
        loc = rng.standard_normal(ls).astype(F32)
        sc = np.exp(0.4 * rng.standard_normal(ss)).astype(F32)
        x = (3.0 * rng.standard_normal(gs)).astype(F32)
        if case == 0:
            x.flat[:4] = [30.0, -30.0, 80.0, -80.0]      # softplus threshold / saturation
        loc_t, sc_t, x_t = t(loc, True), t(sc, True), t(x, True)
        d = Logistic(loc_t, sc_t, group_ndims=g)
        lp = d.log_prob(x_t)
        w = rng.standard_normal(tuple(lp.shape)).astype(F32)
        gl, gsc, gx = torch.autograd.grad((lp * t(w)).sum(), [loc_t, sc_t, x_t])
        p = "c%03d_" % case
        out[p + "loc"], out[p + "scale"], out[p + "x"], out[p + "g"] = loc, sc, x, np.array(g)
        out[p + "lp"], out[p + "w"] = lp, w
        out[p + "gloc"], out[p + "gscale"], out[p + "gx"] = gl, gsc, gx
        case += 1
    out["n_cases"] = np.array(case)
    # scipy-checked value of the reference's own test (test_logistic.py:69-77)
    out["kat_lp"] = Logistic([2.], [1.]).log_prob([3.])
    save("g_logistic_logprob", **out)


def gen_uniform():
    from zhusuan.distributions import Uniform
    rng = np.random.RandomState(808)
    out = {}
    case = 0
    for shape in [(8, 4), (3, 5, 8)]:
        for K in [None, 1, 5]:
            for reparam in [True, False]:
                low = rng.standard_normal(shape).astype(F32)
                high = (low + np.exp(0.5 * rng.standard_normal(shape))).astype(F32)
                ushape = shape if (K is None or K == 1) else (K,) + shape
                u = rng.uniform(0.0, 1.0, ushape).astype(F32)
                low_t, high_t = t(low, True), t(high, True)
                d = Uniform(low_t, high_t, is_reparameterized=reparam)
                with UniformQueue([u]) as q:
                    z = d.sample(K)
                wz = rng.standard_normal(tuple(z.shape)).astype(F32)
                glo, ghi = torch.autograd.grad((z * t(wz)).sum(), [low_t, high_t])
                p = "c%03d_" % case
                out[p + "low"], out[p + "high"], out[p + "u"] = low, high, u
                out[p + "K"] = np.array(-1 if K is None else K)
                out[p + "reparam"] = np.array(int(reparam))
                out[p + "z"], out[p + "cache"], out[p + "wz"] = z, d.sample_cache, wz
                out[p + "glow"], out[p + "ghigh"] = glo, ghi
                out[p + "draw_shape"] = np.array(q.calls[0])
                case += 1
    out["n_cases"] = np.array(case)
    # low-shaped draw shared along high-only broadcast axes (uniform.py:52-53,66-67)
    low = rng.standard_normal((1, 3)).astype(F32)
    high = (low.max() + np.exp(rng.standard_normal((2, 1)))).astype(F32)
    u = rng.uniform(size=(2, 1, 3)).astype(F32)
    d = Uniform(t(low), t(high))
    with UniformQueue([u]) as q:
        z = d.sample(2)
    out["bc_low"], out["bc_high"], out["bc_u"], out["bc_z"] = low, high, u, z
    out["bc_draw_shape"] = np.array(q.calls[0])
    save("g_uniform_sample", **out)

    out = {}
    case = 0
    specs = [((6, 4), (6, 4), (6, 4), 0), ((6, 4), (6, 4), (3, 6, 4), 0), ((6, 4), (6, 4), (3, 6, 4), 1),
             ((5, 7), (5, 7), (4, 5, 7), 2), ((1, 3), (2, 1), (2, 3), 0), ((1,), (1,), (9,), 0)]
    for ls, hs, gs, g in specs:
        low = rng.standard_normal(ls).astype(F32)
        high = (low.max() + np.exp(0.5 * rng.standard_normal(hs))).astype(F32)
        full = np.broadcast_shapes(ls, hs, gs)
        lo_b, hi_b = np.broadcast_to(low, full), np.broadcast_to(high, full)
        # inside the support (the reference's torch.distributions validation raises outside it)
        x_full = (lo_b + rng.uniform(0.05, 0.95, full) * (hi_b - lo_b)).astype(F32)
        x = x_full if tuple(gs) == tuple(full) else x_full[(0,) * (len(full) - len(gs))]
        if tuple(x.shape) != tuple(gs):
            x = np.ascontiguousarray(np.broadcast_to(x_full.min(axis=tuple(range(len(full) - len(gs)))), gs))
            x = np.maximum(x, lo_b.max() + 1e-3).astype(F32)
        low_t, high_t = t(low, True), t(high, True)
        d = Uniform(low_t, high_t, group_ndims=g)
        lp = d.log_prob(t(x))
        w = rng.standard_normal(tuple(lp.shape)).astype(F32)
        glo, ghi = torch.autograd.grad((lp * t(w)).sum(), [low_t, high_t])
        p = "c%03d_" % case
        out[p + "low"], out[p + "high"], out[p + "x"], out[p + "g"] = low, high, x, np.array(g)
        out[p + "lp"], out[p + "w"], out[p + "glow"], out[p + "ghigh"] = lp, w, glo, ghi
        case += 1
    out["n_cases"] = np.array(case)
    out["kat_lp"] = Uniform(np.array([4.]), np.array([5.])).log_prob([4.5])   # test_uniform.py:80 (float64)
    save("g_uniform_logprob", **out)


# --------------------------------------------------------------------------
# G-ST: StochasticTensor.log_prob reductions; G-LME
# --------------------------------------------------------------------------
class _Net(BayesianNet):
    def forward(self, observed):
        self.observe(observed)
        return self


def gen_stochastic_tensor():
    rng = np.random.RandomState(303)
    out = {}
    case = 0
    combos = [
        # (param shape, K, group_ndims, reduce_mean_dims, reduce_sum_dims, multiplier)
        ((6, 4), None, 0, [0], [1], None),          # VAE latent (vae_mnist.py:79-85)
        ((6, 4), 5, 0, None, [2], None),            # IWAE latent (iwae.py:114-119)
        ((5, 7), 4, 2, [0], None, None),            # BNN weights (bnn_vi.py:91-98)
        ((6, 4), 3, 0, [0, 1], None, 456),          # BNN likelihood-style (bnn_vi.py:55-60)
        ((6, 4), 3, 1, None, None, None),
        ((6, 4), 3, 0, None, None, 2.5),
        ((6, 4), 3, 0, [1], [0], None),
        ((2, 6, 4), 3, 0, [1], [3, 2], None),
        ((6, 4), None, 0, None, [-1], None),
    ]
    for shape, K, g, rm, rs, mult in combos:
        mu = rng.standard_normal(shape).astype(F32)
        sd = np.exp(0.3 * rng.standard_normal(shape)).astype(F32)
        eshape = shape if K is None else (K,) + shape
        e1 = rng.standard_normal(eshape).astype(F32)
        e2 = rng.standard_normal(eshape).astype(F32)
        net = _Net()
        net({})
        kw = {}
        if rm is not None:
            kw["reduce_mean_dims"] = rm
        if rs is not None:
            kw["reduce_sum_dims"] = rs
        if mult is not None:
            kw["multiplier"] = mult
        with EpsQueue([e1, e2]):
            first = net.normal("z", mean=t(mu), std=t(sd), group_ndims=g, n_samples=K, **kw)
            second = net.nodes["z"].tensor
        lp = net.nodes["z"].log_prob()
        p = "c%03d_" % case
        out[p + "mu"], out[p + "sd"], out[p + "e1"], out[p + "e2"] = mu, sd, e1, e2
        out[p + "K"] = np.array(-1 if K is None else K)
        out[p + "g"] = np.array(g)
        out[p + "rm"] = np.array(rm if rm is not None else [], dtype=np.int64)
        out[p + "rs"] = np.array(rs if rs is not None else [], dtype=np.int64)
        out[p + "mult"] = np.array(0.0 if mult is None else mult)
        out[p + "first"], out[p + "second"], out[p + "lp"] = first, second, lp
        out[p + "lp_shape"] = np.array(tuple(lp.shape), dtype=np.int64)
        case += 1
    out["n_cases"] = np.array(case)
    save("g_stochastic_tensor", **out)

    # log_mean_exp (zhusuan/utils.py:6-21)
    out = {}
    x = np.array([[1., 2.], [3., -1000.]], F32)
    out["a_x"] = x
    out["a_dim0"] = zhusuan.log_mean_exp(t(x), 0)
    out["a_dim1_keep"] = zhusuan.log_mean_exp(t(x), 1, keepdims=True)
    y = (5 * rng.standard_normal((7, 5, 3))).astype(F32)
    out["b_x"] = y
    out["b_dim0"] = zhusuan.log_mean_exp(t(y), 0)
    out["b_dim1"] = zhusuan.log_mean_exp(t(y), 1)
    out["b_dim2_keep"] = zhusuan.log_mean_exp(t(y), 2, keepdims=True)
    save("g_log_mean_exp", **out)


# --------------------------------------------------------------------------
# G-IW / G-ELBO: estimators on raw log-joint tensors
# --------------------------------------------------------------------------
def _iw(estimator, axis=0):
    obj = ImportanceWeightedObjective.__new__(ImportanceWeightedObjective)
    torch.nn.Module.__init__(obj)
    obj._axis = axis
    obj.estimator = estimator
    return obj


def gen_iw():
    rng = np.random.RandomState(404)
    out = {}
    case = 0
    for (K, B) in [(2, 3), (5, 8), (50, 16), (64, 4), (70, 3), (200, 2)]:
        for spread in [1.0, 5.0, 30.0]:
            logp = (-550.0 + spread * rng.standard_normal((K, B))).astype(F32)
            logq = (-50.0 + 0.3 * spread * rng.standard_normal((K, B))).astype(F32)
            p = "c%03d_" % case
            out[p + "logp"], out[p + "logq"] = logp, logq
            for est in ["sgvb", "vimco"]:
                lp_t, lq_t = t(logp, True), t(logq, True)
                obj = _iw(est)
                cost = getattr(obj, est)(lp_t, lq_t, True)
                gp, gq = torch.autograd.grad(cost, [lp_t, lq_t])
                out[p + est + "_cost"] = cost
                out[p + est + "_glogp"], out[p + est + "_glogq"] = gp, gq
                # float64 evaluation of the same reference code, to size the fp32 error
                lp_d = torch.tensor(logp.astype(np.float64), requires_grad=True)
                lq_d = torch.tensor(logq.astype(np.float64), requires_grad=True)
                cost_d = getattr(obj, est)(lp_d, lq_d, True)
                gp_d, gq_d = torch.autograd.grad(cost_d, [lp_d, lq_d])
                out[p + est + "_cost64"] = cost_d
                out[p + est + "_glogp64"], out[p + est + "_glogq64"] = gp_d, gq_d
            out[p + "sgvb_cost_noreduce"] = _iw("sgvb").sgvb(t(logp), t(logq), False)
            out[p + "bound"] = zhusuan.log_mean_exp(t(logp) - t(logq), 0)
            case += 1
    out["n_cases"] = np.array(case)
    # 1-D log_w (test_iw.py:149-174 runs vimco on a 1-D tensor)
    lp1 = (-3.0 + rng.standard_normal(37)).astype(F32)
    lq1 = (-1.0 + 0.5 * rng.standard_normal(37)).astype(F32)
    for est in ["sgvb", "vimco"]:
        a, b = t(lp1, True), t(lq1, True)
        c = getattr(_iw(est), est)(a, b, True)
        ga, gb = torch.autograd.grad(c, [a, b])
        out["d1_" + est + "_cost"], out["d1_" + est + "_glogp"], out["d1_" + est + "_glogq"] = c, ga, gb
    out["d1_logp"], out["d1_logq"] = lp1, lq1
    save("g_iw", **out)

    # config-3 shape, scalars only
    out = {}
    for i, spread in enumerate([1.0, 5.0, 30.0]):
        r2 = np.random.RandomState(4100 + i)
        logp = (-550.0 + spread * r2.standard_normal((50, 256))).astype(F32)
        logq = (-50.0 + 0.3 * spread * r2.standard_normal((50, 256))).astype(F32)
        for est in ["sgvb", "vimco"]:
            a, b = t(logp, True), t(logq, True)
            c = getattr(_iw(est), est)(a, b, True)
            ga, gb = torch.autograd.grad(c, [a, b])
            out["s%d_%s_cost" % (i, est)] = c
            out["s%d_%s_glogp_sum" % (i, est)] = ga.sum()
            out["s%d_%s_glogq_abs_sum" % (i, est)] = gb.abs().sum()
            out["s%d_%s_glogq_row0" % (i, est)] = gb[:, 0]
        out["s%d_bound_mean" % i] = zhusuan.log_mean_exp(t(logp) - t(logq), 0).mean()
        out["s%d_spread" % i] = np.array(spread)
    save("g_iw_c3", **out)


def gen_elbo():
    rng = np.random.RandomState(505)
    out = {}
    e = ELBO.__new__(ELBO)
    torch.nn.Module.__init__(e)
    a = (-500 + 3 * rng.standard_normal((4, 6))).astype(F32)
    b = (-40 + rng.standard_normal((4, 6))).astype(F32)
    out["logp"], out["logq"] = a, b
    out["sgvb_mean"] = e.sgvb(t(a), t(b), True)
    out["sgvb_nomean"] = e.sgvb(t(a), t(b), False)
    out["sgvb_scalar"] = e.sgvb(t(a[0, 0]), t(b[0, 0]), True)
    save("g_elbo_sgvb", **out)


# --------------------------------------------------------------------------
# G-RF: ELBO.reinforce, three consecutive steps each (the moving mean is state)
# --------------------------------------------------------------------------
def gen_reinforce():
    rng = np.random.RandomState(606)
    out = {}
    case = 0
    specs = [  # (shape, baseline kind, variance_reduction, reduce_mean)
        ((), None, False, True), ((1,), None, True, False), ((7,), None, True, True), ((4, 6), None, True, True),
        ((7,), "tensor", True, True), ((7,), "scalar", True, True), ((1,), "scalar", True, False),
        ((7,), "tensor", False, True), ((7,), None, False, False), ((300,), "tensor", True, True),
    ]
    # (a 0-d log-joint with variance reduction raises in the reference: `l_signal -= moving_mean` cannot broadcast a
    #  [1] buffer into a 0-d tensor in place, elbo.py:225; so does a vector without reduce_mean, elbo.py:221)
    for shape, bkind, vr, rm in specs:
        e = ELBO(None, None, estimator="reinforce")
        p = "c%03d_" % case
        out[p + "vr"], out[p + "rm"] = np.array(int(vr)), np.array(int(rm))
        out[p + "bkind"] = np.array({None: 0, "tensor": 1, "scalar": 2}[bkind])
        for step in range(3):
            a = (-90 + 3 * rng.standard_normal(shape)).astype(F32)
            b = (-40 + rng.standard_normal(shape)).astype(F32)
            a_t, b_t = t(a, True), t(b, True)
            base_t = None
            if bkind == "tensor":
                base = (-50 + rng.standard_normal(shape)).astype(F32)
                base_t = t(base, True)
            elif bkind == "scalar":
                base = np.asarray(-50 + rng.standard_normal(), F32)
                base_t = t(base, True)
            res = e.reinforce(a_t, b_t, reduce_mean=rm, baseline=base_t, variance_reduction=vr, decay=0.8)
            q = p + "s%d_" % step
            out[q + "logp"], out[q + "logq"] = a, b
            if isinstance(res, tuple):
                loss, elbo_mean = res
                out[q + "elbo_mean"] = elbo_mean
            else:
                loss = res
            w = rng.standard_normal(tuple(loss.shape)).astype(F32)
            inputs = [a_t, b_t] + ([base_t] if (base_t is not None and vr) else [])
            grads = torch.autograd.grad((loss * t(w)).sum(), inputs, allow_unused=True)
            out[q + "loss"], out[q + "w"] = loss, w
            out[q + "glogp"] = grads[0] if grads[0] is not None else np.zeros(shape, F32)
            out[q + "glogq"] = grads[1]
            if base_t is not None:
                out[q + "baseline"] = base
                if vr:
                    out[q + "gbaseline"] = grads[2]
            out[q + "moving_mean"] = e.moving_mean.clone()
            out[q + "local_step"] = e.local_step.clone()
        case += 1
    out["n_cases"] = np.array(case)
    save("g_elbo_reinforce", **out)


# --------------------------------------------------------------------------
# End-to-end callers
# --------------------------------------------------------------------------
def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fill_params(module, seed):
    """Deterministic numpy weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); shared with the tests."""
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            shp = tuple(p.shape)
            fan_in = shp[-1] if len(shp) > 1 else shp[0]
            s = 1.0 / np.sqrt(max(fan_in, 1))
            p.copy_(torch.tensor(rng.uniform(-s, s, size=shp).astype(F32)))


def grad_stats(module):
    names, norms, sums, heads = [], [], [], []
    for name, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(name)
        norms.append(float(g.double().norm()))
        sums.append(float(g.double().sum()))
        heads.append(g.reshape(-1)[:8].detach().numpy().copy())
    return names, np.array(norms), np.array(sums), heads


GRAD_FULL_MAX = 32768      # gradients up to this many elements are stored whole, larger ones every GRAD_STRIDE-th element
GRAD_STRIDE = 97           # (prime: the sample walks through every row and column residue of the weight matrices)


def _pack_grads(out, prefix, module, elementwise=False, stride=None, tag=""):
    """Per-parameter gradient norms / sums / first 8 elements; with `elementwise` also the gradient itself ("gfull_*")
    or, for the large weight matrices, a strided sample of it ("gstride_*"), so that the tests can compare element by
    element (a norm cannot see a sign or a permutation error inside a tensor)."""
    names, norms, sums, heads = grad_stats(module)
    out[prefix + "grad_names"] = np.array(names)
    out[prefix + "grad_norms" + tag] = norms
    out[prefix + "grad_sums" + tag] = sums
    for n, h in zip(names, heads):
        out[prefix + "ghead%s_" % tag + n] = h
    if elementwise or stride:
        absmax = []
        for name, p in module.named_parameters():
            g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1).numpy()
            absmax.append(float(np.abs(g).max()))
            if g.size <= (GRAD_FULL_MAX if elementwise else 1024):
                out[prefix + "gfull%s_" % tag + name] = g.copy()
            else:
                out[prefix + "gstride%s_" % tag + name] = g[::(GRAD_STRIDE if elementwise else stride)].copy()
        out[prefix + "grad_absmax" + tag] = np.array(absmax)


def gen_vae(philox=False):
    vae = _load(os.path.join(REF, "examples/variational_autoencoder/vae_mnist.py"), "ref_vae")
    for tag, B, full in [("small", 8, True), ("c1", 64, False), ("c2", 512, False)]:
        if philox and tag == "c1":
            continue
        rng = np.random.RandomState(600 + B)
        x_dim, z_dim = 784, 40
        gen = vae.Generator(x_dim, z_dim, B)
        var = vae.Variational(x_dim, z_dim, B)
        model = ELBO(gen, var)
        fill_params(model, 1000 + B)
        x = (rng.uniform(size=(B, x_dim)) < 0.5).astype(F32)
        e1 = rng.standard_normal((B, z_dim)).astype(F32)
        e2 = rng.standard_normal((B, z_dim)).astype(F32)
        if philox:
            e1 = philox_epsilon(1, B * z_dim, z_dim, PHILOX_SEED, 0).reshape(B, z_dim)
            e2 = philox_epsilon(1, B * z_dim, z_dim, PHILOX_SEED, 1).reshape(B, z_dim)
        with EpsQueue([e1, e2]) as q:
            loss = model({"x": t(x)})
        model.zero_grad()
        loss.backward()
        out = {"B": np.array(B), "seed_params": np.array(1000 + B), "seed_data": np.array(600 + B),
               "loss": loss, "draws": np.array([c[1] for c in q.calls])}
        if philox:
            out["philox_seed"] = np.array(PHILOX_SEED)
        out["logpz"] = gen.nodes["z"].log_prob()
        out["logpx"] = gen.nodes["x"].log_prob()
        out["logqz"] = var.nodes["z"].log_prob()
        _pack_grads(out, "", model, elementwise=full)
        if full:
            out["x"], out["e1"], out["e2"] = x, e1, e2
            out["z"] = var.nodes["z"].dist.sample_cache
            out["x_mean"] = gen.cache["x_mean"]
        save("g_vae_" + tag + ("_philox" if philox else ""), **out)


def gen_vae_philox():
    gen_vae(philox=True)


GRAD_STRIDE_BIG = 997      # config-shape goldens: every 997th element of every gradient


PHILOX_SEED = 20240229     # gen_iwae_philox: the seed of the Philox4x32-10 stream the draws are taken from


def philox_epsilon(K, M, D, seed, call):
    """[K, M] standard-normal draws of the package's sampling kernel for Philox (seed, call id): the C oracle's restatement of
    that kernel (oracle/zs_oracle_c.c, pinned bit for bit against the HIP kernel by tests/test_cabi.py) asked for
    z = 0 + 1 * eps.  Test infrastructure generating INPUTS for the reference; nothing of the oracle's arithmetic beyond the
    stream itself enters the fixture."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(OUT_SRC)), "oracle", "_build", "libzs_oracle.so"))
    fn = lib.zs_normal_sample_logprob_f32
    P, U, L = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64
    fn.argtypes = [P, P, P, U, U, P, P, P, L, L, L, L, L, ctypes.c_int, P, P]
    fn.restype = ctypes.c_int
    mu, sigma = np.zeros(M, F32), np.ones(M, F32)
    z, lp = np.empty((K, M), F32), np.empty((K, M // D), F32)
    rc = fn(mu.ctypes.data, sigma.ctypes.data, None, seed, call, None, z.ctypes.data, lp.ctypes.data, K, M, D, M // D, 1, 0, None, None)
    assert rc == 0, rc
    return z


def gen_iwae(philox=False):
    iw = _load(os.path.join(REF, "examples/variational_autoencoder/iwae.py"), "ref_iwae")
    for est in ["sgvb", "vimco"]:
        # c4g: the GLOBAL batch of BASELINE config 4 (8 GPUs x 256) evaluated by the reference in one process (both estimators)
        for tag, B, K, hidden, full in [("small", 8, 5, 32, True), ("c3", 256, 50, 500, False), ("c4g", 2048, 50, 500, False)]:
            if philox and tag == "c4g":
                continue
            out = {}
            # Two passes of the SAME reference code: float32 (the parity target) and float64 (torch default dtype
            # switched, identical weights / data / epsilon values).  VIMCO's learning signal subtracts two ~|log w|-sized
            # numbers, so the reference's own fp32 gradients are only ~1e-4..1e-3-accurate; the
            # float64 run tells the tests how far a correct fp32 implementation may sit from the fp32 reference.
            for dt, sfx in [(torch.float32, ""), (torch.float64, "64")]:
                torch.set_default_dtype(dt)
                try:
                    iw.hidden_dim = hidden
                    iw.reparameterization = (est == "sgvb")
                    rng = np.random.RandomState(700 + B + K)
                    x_dim, z_dim = 784, 40
                    gen = iw.Generator(x_dim, z_dim, K)
                    var = iw.Variational(x_dim, z_dim, K)
                    model = ImportanceWeightedObjective(gen, var, axis=0, estimator=est)
                    fill_params(model, 2000 + B + K)
                    x = (rng.uniform(size=(B, x_dim)) < 0.5).astype(F32)
                    e1 = rng.standard_normal((K, B, z_dim)).astype(F32)
                    e2 = rng.standard_normal((K, B, z_dim)).astype(F32)
                    if philox:   # the two draws of the latent as the package's kernel makes them for (seed, call ids 0 and 1)
                        e1 = philox_epsilon(K, B * z_dim, z_dim, PHILOX_SEED, 0).reshape(K, B, z_dim)
                        e2 = philox_epsilon(K, B * z_dim, z_dim, PHILOX_SEED, 1).reshape(K, B, z_dim)
                    with EpsQueue([torch.tensor(e1, dtype=dt), torch.tensor(e2, dtype=dt)]) as q:
                        loss = model({"x": torch.tensor(x, dtype=dt)})
                    assert loss.dtype == dt
                    model.zero_grad()
                    loss.backward()
                    logpxz = gen.nodes["z"].log_prob() + gen.nodes["x"].log_prob()
                    logqz = var.nodes["z"].log_prob()
                    log_w = (logpxz - logqz).detach()
                    out["loss" + sfx] = loss
                    out["iw_bound" + sfx] = zhusuan.log_mean_exp(log_w, 0).mean()
                    _pack_grads(out, "", model, elementwise=full, stride=None if full else GRAD_STRIDE_BIG, tag=sfx)
                    if sfx == "64":
                        continue
                    out.update({"B": np.array(B), "K": np.array(K), "hidden": np.array(hidden),
                                "seed_params": np.array(2000 + B + K), "seed_data": np.array(700 + B + K),
                                "draws": np.array([c[1] for c in q.calls]),
                                "draw_kinds": np.array([c[0] for c in q.calls])})
                    if full:
                        out["x"], out["e1"], out["e2"] = x, e1, e2
                        out["z"] = var.nodes["z"].dist.sample_cache
                    if tag != "c4g":  # every log-importance-weight and its three terms ([K, B]: 51 KB each at the C3 shape)
                        out["log_w"] = log_w
                        out["logqz"] = logqz
                        out["logpz"] = gen.nodes["z"].log_prob()
                        out["logpx"] = gen.nodes["x"].log_prob()
                    else:             # global batch: scalars, gradient statistics and a thin slice of log w
                        out["log_w_col0"] = log_w[:, 0]
                        out["log_w_row0_every16"] = log_w[0, ::16]
                        out["bound_b_every16"] = zhusuan.log_mean_exp(log_w, 0)[::16]
                finally:
                    torch.set_default_dtype(torch.float32)
            if philox:
                out["philox_seed"] = np.array(PHILOX_SEED)
            save("g_iwae_%s_%s%s" % (est, tag, "_philox" if philox else ""), **out)


def gen_iwae_philox():
    """The IWAE goldens again with the reference's two draws of the latent taken from the Philox stream the package's OWN
    sampling kernel produces for (PHILOX_SEED, call ids 0 and 1): the product run with that seed and NO injected draw (the
    in-kernel Philox path, both draws in one launch) must reproduce these reference outputs directly."""
    gen_iwae(philox=True)


def gen_bnn(philox=False):
    """philox: the reference's four draws (w0#1, w1#1, w0#2, w1#2) carry the Philox call ids 0, 1, 2, 3 -- ids in the order of
    the draws, which is how the package numbers them for this model (one id per draw of each node, pass by pass)."""
    bnn = _load(os.path.join(REF, "examples/bayesian_neural_nets/bnn_vi.py"), "ref_bnn")
    # c5g: the GLOBAL batch of BASELINE config 5 (8 GPUs x 512) evaluated by the reference in one process
    for tag, B, K, full in [("small", 16, 4, True), ("c5", 512, 10, False), ("c5g", 4096, 10, False)]:
        if philox and tag == "c5g":
            continue
        rng = np.random.RandomState(800 + B + K)
        layer_sizes = [13, 50, 1]
        net = bnn.Net(layer_sizes, K)
        var = bnn.Variational(layer_sizes, K)
        model = ELBO(net, var)
        # non-trivial variational parameters (the example starts from zeros, bnn_vi.py:74-79)
        prng = np.random.RandomState(3000 + B + K)
        with torch.no_grad():
            for p in var.w_means:
                p.copy_(torch.tensor((0.1 * prng.standard_normal(tuple(p.shape))).astype(F32)))
            for p in var.w_logstds:
                p.copy_(torch.tensor((-1.0 + 0.1 * prng.standard_normal(tuple(p.shape))).astype(F32)))
            net.y_logstd.fill_(0.3)
        x = rng.standard_normal((B, 13)).astype(F32)
        y = rng.standard_normal((B,)).astype(F32)
        eps = []
        for _ in range(2):  # draw order w0#1, w1#1, then w0#2, w1#2
            eps.append(rng.standard_normal((K, 50, 14)).astype(F32))
            eps.append(rng.standard_normal((K, 1, 51)).astype(F32))
        if philox:
            ids = (0, 1, 2, 3)
            eps = [philox_epsilon(K, int(np.prod(e.shape[1:])), e.shape[-1], PHILOX_SEED, c).reshape(e.shape) for e, c in zip(eps, ids)]
        with EpsQueue(list(eps)) as q:
            loss = model({"x": t(x), "y": t(y)})
        model.zero_grad()
        loss.backward()
        out = {"B": np.array(B), "K": np.array(K), "loss": loss, "rmse": net.cache["rmse"],
               "seed_params": np.array(3000 + B + K), "seed_data": np.array(800 + B + K),
               "draws_w0": np.array(q.calls[0][1]), "draws_w1": np.array(q.calls[1][1]),
               "n_draws": np.array(len(q.calls))}
        for i in range(2):
            out["g_w_mean_%d" % i] = var.w_means[i].grad
            out["g_w_logstd_%d" % i] = var.w_logstds[i].grad
        out["g_y_logstd"] = net.y_logstd.grad
        out["logp_w0"] = net.nodes["w0"].log_prob()
        out["logp_w1"] = net.nodes["w1"].log_prob()
        out["logp_y"] = net.nodes["y"].log_prob()
        out["logq_w0"] = var.nodes["w0"].log_prob()
        out["logq_w1"] = var.nodes["w1"].log_prob()
        if full:
            out["x"], out["y"] = x, y
            for i, e in enumerate(eps):
                out["eps%d" % i] = e
        if philox:
            out["philox_seed"], out["philox_call_ids"] = np.array(PHILOX_SEED), np.array(ids)
            if full:
                for i in range(4):
                    del out["eps%d" % i]
        save("g_bnn_" + tag + ("_philox" if philox else ""), **out)


def gen_bnn_philox():
    gen_bnn(philox=True)


def gen_uniform_latent():
    """A non-reparameterised Uniform latent under VIMCO and under REINFORCE: the reference draws it without
    gradient but rescales the draw OUTSIDE the no-grad region (uniform.py:63-70), so the value handed to the generator
    carries d/d low = 1 - u, d/d high = u and the generator's log-joint back-propagates into low / high.  Also pins the
    reference's quirk that sample_cache (the value log q is evaluated at) is the once-scaled draw while the returned
    sample is scaled twice."""
    from zhusuan.distributions import Uniform
    B, K, D = 6, 4, 3

    class Q(BayesianNet):
        def __init__(self):
            super().__init__()
            self.low = torch.nn.Parameter(torch.zeros(D))
            self.logw = torch.nn.Parameter(torch.zeros(D))

        def forward(self, observed):
            self.observe(observed)
            low = self.low.unsqueeze(0).expand(B, D)
            high = low + torch.exp(self.logw).unsqueeze(0).expand(B, D)
            self.sn(Uniform(low, high, is_reparameterized=False), "z", n_samples=K, reduce_sum_dims=[2])
            return self

    class P(BayesianNet):
        def __init__(self):
            super().__init__()
            self.scale = torch.nn.Parameter(torch.ones(D))

        def forward(self, observed):
            self.observe(observed)
            z = self.normal("z", mean=torch.zeros(B, D), std=3. * torch.ones(B, D), n_samples=K, reduce_sum_dims=[2])
            self.normal("x", mean=z * self.scale, std=torch.ones(B, D), reduce_sum_dims=[2])
            return self

    rng = np.random.RandomState(909)
    out = {"B": np.array(B), "K": np.array(K), "D": np.array(D)}
    low0 = (0.3 * rng.standard_normal(D)).astype(F32)
    logw0 = (0.2 * rng.standard_normal(D)).astype(F32)
    scale0 = (1.0 + 0.1 * rng.standard_normal(D)).astype(F32)
    x = rng.standard_normal((B, D)).astype(F32)
    u1 = rng.uniform(0.02, 0.98, (K, B, D)).astype(F32)
    u2 = rng.uniform(0.02, 0.98, (K, B, D)).astype(F32)
    out.update(low=low0, logw=logw0, scale=scale0, x=x, u1=u1, u2=u2)
    for est in ["vimco", "reinforce"]:
        q, pnet = Q(), P()
        with torch.no_grad():
            q.low.copy_(t(low0)); q.logw.copy_(t(logw0)); pnet.scale.copy_(t(scale0))
        if est == "vimco":
            model = ImportanceWeightedObjective(pnet, q, axis=0, estimator="vimco")
        else:
            model = ELBO(pnet, q, estimator="reinforce")
        with UniformQueue([u1, u2]) as uq:
            res = model({"x": t(x)})
        loss = res[0] if isinstance(res, tuple) else res
        model.zero_grad()
        loss.backward()
        out[est + "_loss"] = loss
        out[est + "_g_low"], out[est + "_g_logw"], out[est + "_g_scale"] = q.low.grad, q.logw.grad, pnet.scale.grad
        out[est + "_z_used"] = pnet.nodes["z"].tensor if False else pnet.observed["z"]
        out[est + "_logq"] = q.nodes["z"].log_prob()
        out[est + "_draws"] = np.array([list(c) for c in uq.calls])
    save("g_uniform_latent", **out)


def gen_seeded():
    """The reference's example models run from ``torch.manual_seed`` with NO draw injected: what a user's seeded run of
    the reference produces.  The build's seed-compatible mode (``zhusuan.reference_rng()``: host draws from the same CPU
    stream, call for call) must reproduce these numbers -- which also pins the draw ORDER (two draws per latent, the
    second one used; the BNN's w0#1, w1#1, w0#2, w1#2)."""
    out = {}
    seed = 4321
    vae = _load(os.path.join(REF, "examples/variational_autoencoder/vae_mnist.py"), "ref_vae_s")
    B = 8
    rng = np.random.RandomState(600 + B)
    model = ELBO(vae.Generator(784, 40, B), vae.Variational(784, 40, B))
    fill_params(model, 1000 + B)
    x = (rng.uniform(size=(B, 784)) < 0.5).astype(F32)
    torch.manual_seed(seed)
    loss = model({"x": t(x)})
    model.zero_grad()
    loss.backward()
    out["vae_loss"], out["vae_z"] = loss, model.variational.nodes["z"].dist.sample_cache
    out["vae_grad_norms"] = grad_stats(model)[1]
    iw = _load(os.path.join(REF, "examples/variational_autoencoder/iwae.py"), "ref_iwae_s")
    B, K, hidden = 8, 5, 32
    for est in ["sgvb", "vimco"]:
        iw.hidden_dim = hidden
        iw.reparameterization = (est == "sgvb")
        rng = np.random.RandomState(700 + B + K)
        model = ImportanceWeightedObjective(iw.Generator(784, 40, K), iw.Variational(784, 40, K), axis=0, estimator=est)
        fill_params(model, 2000 + B + K)
        x = (rng.uniform(size=(B, 784)) < 0.5).astype(F32)
        torch.manual_seed(seed)
        loss = model({"x": t(x)})
        model.zero_grad()
        loss.backward()
        out["iwae_%s_loss" % est], out["iwae_%s_z" % est] = loss, model.variational.nodes["z"].dist.sample_cache
        out["iwae_%s_grad_norms" % est] = grad_stats(model)[1]
    bnn = _load(os.path.join(REF, "examples/bayesian_neural_nets/bnn_vi.py"), "ref_bnn_s")
    B, K = 16, 4
    rng = np.random.RandomState(800 + B + K)
    net, var = bnn.Net([13, 50, 1], K), bnn.Variational([13, 50, 1], K)
    model = ELBO(net, var)
    prng = np.random.RandomState(3000 + B + K)
    with torch.no_grad():
        for p in var.w_means:
            p.copy_(torch.tensor((0.1 * prng.standard_normal(tuple(p.shape))).astype(F32)))
        for p in var.w_logstds:
            p.copy_(torch.tensor((-1.0 + 0.1 * prng.standard_normal(tuple(p.shape))).astype(F32)))
        net.y_logstd.fill_(0.3)
    x = rng.standard_normal((B, 13)).astype(F32)
    y = rng.standard_normal((B,)).astype(F32)
    torch.manual_seed(seed)
    loss = model({"x": t(x), "y": t(y)})
    model.zero_grad()
    loss.backward()
    out["bnn_loss"] = loss
    out["bnn_g_w_mean_0"], out["bnn_g_w_logstd_1"] = var.w_means[0].grad, var.w_logstds[1].grad
    out["seed"] = np.array(seed)
    save("g_seeded", **out)


def gen_reference_tests():
    """Values of the reference's own statistical tests (test/variational/test_elbo.py, test_iw.py)
    so the build can re-run them against identical expectations."""
    from scipy import stats
    out = {}
    rng = np.random.RandomState(1)
    n1 = rng.standard_normal(size=(1, 1000)).astype(F32)
    n3 = rng.standard_normal(10000).astype(F32)
    # test_iw.py:149-174 (vimco grads vs sgvb grads) for the two parameterisations
    for tag, xm, xs in [("a", 0., 1.), ("b", 2., 3.)]:
        mu = torch.tensor(2., requires_grad=True)
        sigma = torch.tensor(3., requires_grad=True)
        eps = torch.tensor(n3)
        qx = eps * sigma + mu
        norm = Normal(mean=mu, std=sigma)
        log_qx = norm.log_prob(qx)
        vq = eps * sigma.detach() + mu.detach()
        vlog = norm.log_prob(vq)
        pm, ps = torch.tensor(xm), torch.tensor(xs)
        lp_s = Normal(mean=pm, std=ps).log_prob(qx)
        lp_v = Normal(mean=pm, std=ps).log_prob(vq)
        cs = _iw("sgvb").sgvb(lp_s, log_qx, True)
        cv = _iw("vimco").vimco(lp_v, vlog, True)
        gs = torch.autograd.grad(cs, [mu, sigma], retain_graph=True)
        gv = torch.autograd.grad(cv, [mu, sigma], retain_graph=True)
        out[tag + "_sgvb_cost"], out[tag + "_vimco_cost"] = cs, cv
        out[tag + "_sgvb_grads"] = torch.stack(gs)
        out[tag + "_vimco_grads"] = torch.stack(gv)
    out["n1_head"] = n1[0, :8]
    out["n3_head"] = n3[:8]
    save("g_reference_tests", **out)


def gen_error_conventions():
    """Where the reference RAISES (and what) on unusual layouts -- the build must raise the same exception type there and
    agree on the value everywhere else (SURVEY.md section 8b "error conventions"):
      * ImportanceWeightedObjective.sgvb / .vimco on 1-D / 2-D / 3-D log-weights with every axis
        (importance_weighted_objective.py:123-132,152-191: VIMCO only works for 1-D and [K, B] with axis=0);
      * log_prob of a value with 0, 1, 2 or 3 more leading axes than the parameters for the four hand-written families
        (normal.py:112-116, bernoulli.py:88-90, logistic.py:73-77, uniform.py:73-77: the parameters are repeated
        x.shape[0] times, which only lines up for ONE extra axis or equal leading sizes)."""
    from zhusuan.distributions import Logistic, Uniform
    rng = np.random.RandomState(77)
    out = {}
    names, outcomes = [], []

    def record(name, fn):
        try:
            v = fn()
            outcomes.append("ok")
            out[name + "_value"] = v
        except Exception as e:                       # noqa: BLE001  (the TYPE is the datum)
            outcomes.append(type(e).__name__)
        names.append(name)

    i = 0
    for shape in [(7,), (3, 5), (4, 4), (4, 3, 5)]:
        logp = (-20.0 + 2.0 * rng.standard_normal(shape)).astype(F32)
        logq = (-5.0 + rng.standard_normal(shape)).astype(F32)
        for axis in range(-len(shape) - 1, len(shape) + 1):
            for est in ["sgvb", "vimco"]:
                for reduce_mean in [True, False]:
                    name = "iw%03d" % i
                    out[name + "_logp"], out[name + "_logq"] = logp, logq
                    out[name + "_axis"], out[name + "_reduce_mean"] = np.array(axis), np.array(reduce_mean)
                    out[name + "_est"] = np.array(est)
                    record(name, lambda: getattr(_iw(est, axis), est)(t(logp), t(logq), reduce_mean))
                    i += 1
    out["n_iw"] = np.array(i)
    par_a = (0.3 * rng.standard_normal((3, 4))).astype(F32)
    par_b = rng.uniform(0.5, 1.5, size=(3, 4)).astype(F32)
    fams = {
        "normal": lambda: Normal(mean=t(par_a), std=t(par_b)),
        "bernoulli": lambda: Bernoulli(probs=t(par_b / 2.0)),
        "logistic": lambda: Logistic(loc=t(par_a), scale=t(par_b)),
        "uniform": lambda: Uniform(low=t(par_a - 3.0), high=t(par_b + 3.0)),
    }
    out["par_a"], out["par_b"] = par_a, par_b
    j = 0
    for fam in ["normal", "bernoulli", "logistic", "uniform"]:
        for xs in [(3, 4), (4,), (1, 4), (2, 3, 4), (2, 1, 4), (5, 2, 3, 4), (5, 5, 3, 4), (1, 2, 3, 4), (2, 1, 3, 4),
                   (6, 5, 2, 3, 4), (2, 2, 2, 3, 4)]:
            x = rng.uniform(0.0, 1.0, size=xs).astype(F32)
            if fam == "bernoulli":
                x = (x < 0.5).astype(F32)
            name = "lp%03d" % j
            out[name + "_x"], out[name + "_family"] = x, np.array(fam)
            record(name, lambda: fams[fam]().log_prob(t(x)))
            j += 1
    out["n_lp"] = np.array(j)
    out["names"], out["outcomes"] = np.array(names), np.array(outcomes)
    save("g_error_conventions", **out)


# --------------------------------------------------------------------------
# G-OBS: gradients w.r.t. a Bernoulli OBSERVATION (bernoulli.py:94 is differentiable in `sample`; `given` keeps its graph
# through base.py:161-178): a model whose observed value comes out of a differentiable net
# --------------------------------------------------------------------------
def gen_observation_grad():
    rng = np.random.RandomState(909)
    out = {}
    case = 0
    # distribution level: x fractional, a leaf that requires a gradient; x [B, X] against p [K, B, X] (the gradient sums over
    # K), x of full size, a row vector x [X]; probs and logits constructors; group_ndims 0 / 1
    for pshape, xshape, g, from_logits in [((3, 6, 16), (6, 16), 1, 0), ((3, 6, 16), (3, 6, 16), 1, 0), ((4, 5, 12), (12,), 0, 0),
                                           ((3, 6, 16), (6, 16), 1, 1), ((2, 5, 784), (5, 784), 1, 0), ((7, 9), (7, 9), 0, 1)]:
        x = rng.uniform(0.02, 0.98, xshape).astype(F32)
        x_t = t(x, True)
        if from_logits:
            par = (2.5 * rng.standard_normal(pshape)).astype(F32)
            par_t = t(par, True)
            d = Bernoulli(logits=par_t, group_ndims=g)
        else:
            par = rng.uniform(0.001, 0.999, pshape).astype(F32)
            par_t = t(par, True)
            d = Bernoulli(probs=par_t, group_ndims=g)
        lp = d.log_prob(x_t)
        w = rng.standard_normal(tuple(lp.shape)).astype(F32)
        gpar, gx = torch.autograd.grad((lp * t(w)).sum(), [par_t, x_t])
        p = "c%03d_" % case
        out[p + "param"], out[p + "x"], out[p + "lp"], out[p + "w"] = par, x, lp, w
        out[p + "gparam"], out[p + "gx"] = gpar, gx
        out[p + "from_logits"], out[p + "g"] = np.array(from_logits), np.array(g)
        case += 1
    out["n_cases"] = np.array(case)
    # objective level, the reference's examples with the observation a leaf that requires a gradient:
    #   IWAE (both estimators; the package's one-launch generator side) and the VAE's scalar ELBO (the one-launch log-joint)
    iw = _load(os.path.join(REF, "examples/variational_autoencoder/iwae.py"), "ref_iwae_obs")
    B, K, hidden, x_dim, z_dim = 8, 5, 32, 784, 40
    for est in ["sgvb", "vimco"]:
        iw.hidden_dim = hidden
        iw.reparameterization = (est == "sgvb")
        r2 = np.random.RandomState(910)
        gen = iw.Generator(x_dim, z_dim, K)
        var = iw.Variational(x_dim, z_dim, K)
        model = ImportanceWeightedObjective(gen, var, axis=0, estimator=est)
        fill_params(model, 2000 + B + K)
        x = r2.uniform(0.02, 0.98, (B, x_dim)).astype(F32)
        e1 = r2.standard_normal((K, B, z_dim)).astype(F32)
        e2 = r2.standard_normal((K, B, z_dim)).astype(F32)
        x_t = t(x, True)
        with EpsQueue([t(e1), t(e2)]):
            loss = model({"x": x_t})
        model.zero_grad()
        loss.backward()
        pre = "iwae_%s_" % est
        out[pre + "x"], out[pre + "e1"], out[pre + "e2"], out[pre + "loss"], out[pre + "gx"] = x, e1, e2, loss, x_t.grad
        _pack_grads(out, pre, model, stride=GRAD_STRIDE)          # (the parameters' gradients: a strided sample; gx is whole)
        out[pre + "shape"] = np.array([B, K, hidden])
    vae = _load(os.path.join(REF, "examples/variational_autoencoder/vae_mnist.py"), "ref_vae_obs")
    B = 8
    r3 = np.random.RandomState(911)
    gen = vae.Generator(x_dim, z_dim, B)
    var = vae.Variational(x_dim, z_dim, B)
    model = ELBO(gen, var)
    fill_params(model, 1000 + B)
    x = r3.uniform(0.02, 0.98, (B, x_dim)).astype(F32)
    e1 = r3.standard_normal((B, z_dim)).astype(F32)
    e2 = r3.standard_normal((B, z_dim)).astype(F32)
    x_t = t(x, True)
    with EpsQueue([e1, e2]):
        loss = model({"x": x_t})
    model.zero_grad()
    loss.backward()
    out["vae_x"], out["vae_e1"], out["vae_e2"], out["vae_loss"], out["vae_gx"] = x, e1, e2, loss, x_t.grad
    _pack_grads(out, "vae_", model, stride=GRAD_STRIDE)
    save("g_observation_grad", **out)


if __name__ == "__main__":
    gen_error_conventions()
    gen_normal()
    gen_bernoulli()
    gen_stochastic_tensor()
    gen_iw()
    gen_elbo()
    gen_vae()
    gen_iwae()
    gen_bnn()
    gen_reference_tests()
    gen_logistic()
    gen_uniform()
    gen_reinforce()
    gen_uniform_latent()
    gen_seeded()
    gen_iwae_philox()
    gen_vae_philox()
    gen_bnn_philox()
    gen_observation_grad()
