"""Size-independent properties of the hot path at BASELINE.json's FULL sizes (no oracle needed at these
sizes): C3 / per-GPU C4 = IWAE B=256 K=50 X=784, C5 per-GPU = BNN B=512 K=10, plus a 16x larger batch.
GPU only."""
import numpy as np
import pytest
import torch

import zhusuan as zs
from zhusuan.distributions import Normal, Bernoulli
from zhusuan.variational.importance_weighted_objective import ImportanceWeightedObjective

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _iw(est):
    return ImportanceWeightedObjective(None, None, axis=0, estimator=est)


@pytest.mark.parametrize("B", [256, 4096])
def test_bernoulli_rowsum_additivity_and_paths(B):
    K, X = 50, 784
    g = torch.Generator(device=DEV).manual_seed(B)
    logits = 3 * torch.randn(K, B, X, device=DEV, generator=g)
    p = torch.sigmoid(logits)
    x = (torch.rand(B, X, device=DEV, generator=g) < 0.5).float()
    full = Bernoulli(probs=p, group_ndims=1).log_prob(x)                         # [K, B] row sums, one kernel
    assert full.shape == (K, B) and full.stride() == (1, K)
    # additivity: splitting the pixel axis and adding the two row sums gives the same totals
    a = Bernoulli(probs=p[..., :400].contiguous(), group_ndims=1).log_prob(x[:, :400].contiguous())
    b = Bernoulli(probs=p[..., 400:].contiguous(), group_ndims=1).log_prob(x[:, 400:].contiguous())
    torch.testing.assert_close(a + b, full, rtol=2e-6, atol=2e-3)
    # element-wise kernel path (D = 1) summed by torch agrees with the fused row sum
    elem = Bernoulli(probs=p).log_prob(x)
    assert elem.shape == (K, B, X)
    torch.testing.assert_close(elem.sum(-1), full, rtol=2e-6, atol=2e-3)
    # logits constructor (sigmoid inside the kernel) == probs constructor
    via_logits = Bernoulli(logits=logits, group_ndims=1).log_prob(x)
    torch.testing.assert_close(via_logits, full, rtol=2e-5, atol=2e-2)
    # float64 evaluation of the reference formula on a slice
    ref = (x.double() * torch.log(p[:3].double() + 1e-8) + (1 - x.double()) * torch.log(1 - p[:3].double() + 1e-8)).sum(-1)
    torch.testing.assert_close(full[:3].double(), ref, rtol=2e-6, atol=2e-3)
    # x in {0,1}: flipping every bit swaps the roles of p and 1-p (exactly representable p only:
    # 1-(1-p) != p in fp32 in general)
    ph = torch.round(p * 64) / 64
    a1 = Bernoulli(probs=ph, group_ndims=1).log_prob(x)
    a2 = Bernoulli(probs=(1 - ph), group_ndims=1).log_prob(1 - x)
    torch.testing.assert_close(a1, a2, rtol=2e-6, atol=2e-3)


@pytest.mark.parametrize("B", [256, 4096])
def test_normal_sample_logprob_properties(B):
    K, D = 50, 40
    g = torch.Generator(device=DEV).manual_seed(7)
    mu = torch.randn(B, D, device=DEV, generator=g)
    sd = torch.exp(0.3 * torch.randn(B, D, device=DEV, generator=g))
    d = Normal(mean=mu, std=sd, group_ndims=1)
    torch.manual_seed(11)
    z = d.sample(K)
    lq = d.log_prob(None)                                     # fused with the draw
    assert z.shape == (K, B, D) and lq.shape == (K, B) and lq.stride() == (1, K)
    # the fused value equals the stand-alone kernel on the same sample, and float64 math
    lq2 = Normal(mean=mu, std=sd, group_ndims=1).log_prob(z.clone())
    torch.testing.assert_close(lq, lq2, rtol=1e-5, atol=1e-4)
    eps = (z - mu) / sd
    ref = (-0.5 * np.log(2 * np.pi) - torch.log(sd.double()) - 0.5 * eps.double() ** 2).sum(-1)
    torch.testing.assert_close(lq.double(), ref, rtol=1e-5, atol=2e-4)
    # standardised draws are N(0,1): moments at 5e5..8e6 samples
    e = eps.flatten()
    assert abs(float(e.mean())) < 5e-3 and abs(float(e.std()) - 1) < 5e-3
    assert abs(float((e ** 3).mean())) < 2e-2 and abs(float((e ** 4).mean()) - 3) < 5e-2
    # location/scale equivariance of the draw for a fixed Philox call id
    torch.manual_seed(11)
    z_shift = Normal(mean=mu + 2.0, std=sd, group_ndims=1).sample(K)
    torch.testing.assert_close(z_shift - 2.0, z, rtol=0, atol=2e-6)
    # prior N(0, I): log p(z) = -0.5*||z||^2 - D/2 log(2 pi)
    lp = Normal(mean=torch.zeros(B, D, device=DEV), std=torch.ones(B, D, device=DEV), group_ndims=1).log_prob(z)
    torch.testing.assert_close(lp, -0.5 * (z ** 2).sum(-1) - 0.5 * D * np.log(2 * np.pi), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,K", [(256, 50), (2048, 50), (512, 10), (64, 64)])
@pytest.mark.parametrize("est", ["sgvb", "vimco"])
def test_iw_reduce_invariances(B, K, est):
    g = torch.Generator(device=DEV).manual_seed(B + K)
    logp = (-550 + 5 * torch.randn(K, B, device=DEV, generator=g)).requires_grad_(True)
    logq = (-50 + 2 * torch.randn(K, B, device=DEV, generator=g)).requires_grad_(True)
    obj = _iw(est)
    cost = getattr(obj, est)(logp, logq, True)
    gp, gq = torch.autograd.grad(cost, [logp, logq])
    bound = obj.last_iw_bound.clone()
    # d cost / d logp = -softmax_k(log w) / B: columns sum to -1/B
    torch.testing.assert_close(gp.sum(0), torch.full([B], -1.0 / B, device=DEV), rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(-gp * B, torch.softmax((logp - logq).detach(), 0), rtol=1e-4, atol=1e-7)
    # IW bound: between the mean and the max of log w, and equal to float64 logsumexp - log K
    lw = (logp - logq).detach()
    assert bool((bound <= lw.max(0).values + 1e-4).all()) and bool((bound >= lw.mean(0) - 1e-4).all())
    torch.testing.assert_close(bound.double(), torch.logsumexp(lw.double(), 0) - np.log(K), rtol=1e-6, atol=1e-4)
    # permuting the particles permutes the gradients and leaves cost / bound unchanged
    perm = torch.randperm(K, device=DEV, generator=g)
    lp2, lq2 = logp.detach()[perm].requires_grad_(True), logq.detach()[perm].requires_grad_(True)
    obj2 = _iw(est)
    cost2 = getattr(obj2, est)(lp2, lq2, True)
    gp2, gq2 = torch.autograd.grad(cost2, [lp2, lq2])
    torch.testing.assert_close(cost2, cost, rtol=2e-6, atol=1e-4)
    torch.testing.assert_close(obj2.last_iw_bound, bound, rtol=1e-6, atol=1e-4)
    torch.testing.assert_close(gq2, gq[perm], rtol=1e-3, atol=2e-6)
    # shift equivariance: adding c to every log p shifts the bound by c and leaves the weights alone
    obj3 = _iw(est)
    getattr(obj3, est)(logp.detach() + 3.0, logq.detach(), True)
    torch.testing.assert_close(obj3.last_iw_bound, bound + 3.0, rtol=1e-6, atol=2e-4)
    if est == "sgvb":
        torch.testing.assert_close(gq, -gp, rtol=0, atol=0)          # log w = log p - log q
    # log_mean_exp agrees with the bound
    torch.testing.assert_close(zs.log_mean_exp(lw, 0), bound, rtol=1e-6, atol=1e-4)


def test_end_to_end_step_is_deterministic_given_seed():
    from examples import iwae
    torch.manual_seed(0)
    model = iwae.build(n_samples=50, estimator="vimco", hidden=500, device=DEV)
    x = (torch.rand(256, 784, device=DEV) < 0.5).float()
    vals = []
    for _ in range(2):
        torch.manual_seed(123)
        loss = model({"x": x})
        model.zero_grad()
        loss.backward()
        vals.append((float(loss), float(sum(p.grad.double().abs().sum() for p in model.parameters()))))
    assert vals[0][0] == vals[1][0]                         # same Philox call ids -> bit-identical objective
    assert abs(vals[0][1] - vals[1][1]) <= 1e-6 * vals[0][1]


# ------------------------------------------------------------------ widened rows at full sizes
@pytest.mark.parametrize("B", [256, 4096])
def test_logistic_uniform_properties(B):
    from scipy import stats
    from zhusuan.distributions import Logistic, Uniform
    K, D = 50, 40
    g = torch.Generator(device=DEV).manual_seed(B + 1)
    loc = torch.randn(B, D, device=DEV, generator=g)
    scale = torch.rand(B, D, device=DEV, generator=g) + 0.5
    d = Logistic(loc, scale, group_ndims=1)
    torch.manual_seed(7)
    z = d.sample(K)
    lp_fused = d.log_prob(None)                                   # L1: density of the fresh sample, same launch
    assert z.shape == (K, B, D) and lp_fused.shape == (K, B) and lp_fused.stride() == (1, K)
    lp_given = d.log_prob(z.clone())                              # L2 on the same values
    torch.testing.assert_close(lp_given, lp_fused, rtol=1e-5, atol=2e-4)
    elem = Logistic(loc, scale).log_prob(z)                       # D = 1 path, summed by torch
    torch.testing.assert_close(elem.sum(-1), lp_fused, rtol=1e-5, atol=2e-4)
    ref = stats.logistic.logpdf(z[:3, :5].double().cpu().numpy(), loc[:5].double().cpu().numpy(), scale[:5].double().cpu().numpy()).sum(-1)
    np.testing.assert_allclose(lp_fused[:3, :5].cpu().numpy(), ref, rtol=2e-5, atol=2e-4)
    # affine equivariance: sampling with the same Philox ids from (loc + c, s * scale) moves the sample accordingly
    torch.manual_seed(7)
    z2 = Logistic(loc + 2.0, 3.0 * scale, group_ndims=1).sample(K)
    torch.testing.assert_close(z2, (z - loc) * 3.0 + loc + 2.0, rtol=2e-5, atol=2e-5)
    # Uniform: inside the support the density is the constant -sum log(high - low); the sample stays inside
    low, high = loc - scale, loc + scale
    u = Uniform(low, high, group_ndims=1)
    s = u.sample(K)
    assert bool(((s >= low) & (s < high)).all())
    lpu = u.log_prob(s)
    torch.testing.assert_close(lpu, (-torch.log(high - low).sum(-1)).expand(K, B), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,K", [(256, 50), (2048, 50), (512, 10)])
@pytest.mark.parametrize("est", ["sgvb", "vimco"])
def test_iw_objective_fused_equals_composed(B, K, est):
    """K4b (one launch: two-term log-joint, batch mean, scaled coefficients) against the same objective composed from
    K4 + torch ops, values and gradients, at the BASELINE sizes."""
    from zhusuan import _ops
    g = torch.Generator(device=DEV).manual_seed(B + K)
    a = (-540 + 5 * torch.randn(B, K, device=DEV, generator=g)).requires_grad_()
    b = (-45 + torch.randn(B, K, device=DEV, generator=g)).requires_grad_()
    q = (-50 + 2 * torch.randn(B, K, device=DEV, generator=g)).requires_grad_()
    code = _ops.ZS_IW_VIMCO if est == "vimco" else _ops.ZS_IW_SGVB
    fused, bound = _ops.IWObjective.apply(a, b, q, code, True)
    gf = torch.autograd.grad(fused, [a, b, q])
    cost_b, bound2 = _ops.IWReduce.apply(a + b, q, code)
    comp = cost_b.mean()
    gc = torch.autograd.grad(comp, [a, b, q])
    torch.testing.assert_close(bound, bound2, rtol=0, atol=0)
    torch.testing.assert_close(fused, comp, rtol=2e-6, atol=0)
    for x, y in zip(gf, gc):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-10)
    # adding a constant to both log p terms' sum and to log q leaves the weights, hence the sgvb gradients, unchanged
    if est == "sgvb":
        fused2, _ = _ops.IWObjective.apply(a + 3.0, b, q + 3.0, code, True)
        g2 = torch.autograd.grad(fused2, [a, q])
        torch.testing.assert_close(g2[0], gf[0], rtol=2e-4, atol=1e-7)
