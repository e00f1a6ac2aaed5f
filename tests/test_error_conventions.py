"""Error conventions of the reference on unusual layouts: where the real reference raises, this
package raises the same exception TYPE; where it returns a value, this package returns the same value.  The fixture
`g_error_conventions` was produced by running the reference itself (tests/golden/gen_golden.py:gen_error_conventions)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from zhusuan.distributions import Normal, Bernoulli, Logistic, Uniform
from zhusuan.variational.importance_weighted_objective import ImportanceWeightedObjective

# ONE deliberate difference: VIMCO on a SQUARE [K, K] log-weight matrix with axis=1.  The reference's permutation is the
# identity there (importance_weighted_objective.py:176-186), so for a non-square matrix its broadcast fails with a
# RuntimeError, and for a square one it silently returns a number that mixes rows and columns.  Refused in both cases.
DELIBERATE = {"vimco square axis=1": "RuntimeError"}


def _iw(est, axis):
    obj = ImportanceWeightedObjective.__new__(ImportanceWeightedObjective)
    torch.nn.Module.__init__(obj)
    obj._axis, obj.estimator, obj.last_iw_bound = axis, est, None
    return obj


def _outcome(fn):
    try:
        return "ok", fn()
    except Exception as e:                       # noqa: BLE001
        return type(e).__name__, None


def test_iw_layouts_raise_like_the_reference(dev):
    g = load_golden("g_error_conventions")
    want = dict(zip(g["names"].tolist(), g["outcomes"].tolist()))
    n_ok = n_raise = 0
    for i in range(int(g["n_iw"])):
        name = "iw%03d" % i
        logp, logq = g[name + "_logp"], g[name + "_logq"]
        axis, est, rm = int(g[name + "_axis"]), str(g[name + "_est"]), bool(g[name + "_reduce_mean"])
        got, val = _outcome(lambda: getattr(_iw(est, axis), est)(torch.tensor(logp, device=dev), torch.tensor(logq, device=dev), rm))
        expect = want[name]
        if est == "vimco" and logp.ndim == 2 and logp.shape[0] == logp.shape[1] and axis == 1:
            assert expect == "ok"                                  # the reference's meaningless number
            expect = DELIBERATE["vimco square axis=1"]
        assert got == expect, (name, est, axis, logp.shape, rm, got, expect)
        if got == "ok":
            ref = g[name + "_value"]
            assert tuple(val.shape) == tuple(ref.shape), (name, tuple(val.shape), ref.shape)
            np.testing.assert_allclose(val.cpu().numpy(), ref, rtol=2e-5, atol=2e-5, err_msg=name)
            n_ok += 1
        else:
            n_raise += 1
    assert n_ok >= 36 and n_raise >= 50


def test_log_prob_of_values_with_extra_leading_axes(dev):
    g = load_golden("g_error_conventions")
    want = dict(zip(g["names"].tolist(), g["outcomes"].tolist()))
    a, b = torch.tensor(g["par_a"], device=dev), torch.tensor(g["par_b"], device=dev)
    fams = {"normal": lambda: Normal(mean=a, std=b), "bernoulli": lambda: Bernoulli(probs=b / 2.0),
            "logistic": lambda: Logistic(loc=a, scale=b), "uniform": lambda: Uniform(low=a - 3.0, high=b + 3.0)}
    raised = 0
    for j in range(int(g["n_lp"])):
        name = "lp%03d" % j
        fam, x = str(g[name + "_family"]), g[name + "_x"]
        got, val = _outcome(lambda: fams[fam]().log_prob(torch.tensor(x, device=dev)))
        expect = want[name]
        assert got == expect, (name, fam, x.shape, got, expect)
        if got == "ok":
            ref = g[name + "_value"]
            assert tuple(val.shape) == tuple(ref.shape), (name, tuple(val.shape), ref.shape)
            np.testing.assert_allclose(val.cpu().numpy(), ref, rtol=2e-5, atol=2e-6, err_msg=name)
        else:
            raised += 1
    assert raised == 8
