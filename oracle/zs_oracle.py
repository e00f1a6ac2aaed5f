"""CPU oracle for the variational-inference hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU (fp32) restatement of what the reference
(thuwzy/ZhuSuan-PyTorch, mounted at /root/reference in the build container)
computes on the path named by BASELINE.json:north_star.  It exists so that the
HIP kernels can be checked against it; it is NOT part of the product:

  * only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg
    may import it;
  * the shipped package (zhusuan-pytorch_amd/zhusuan) never imports or calls it
    and fails loudly when its HIP library is missing.

Parity is PINNED: tests/test_oracle_golden.py checks every function below
against the fixtures in tests/golden/*.npz, which were produced by running the
real reference (tests/golden/gen_golden.py) in the build container.

Each function cites the reference file:line whose op sequence it restates
(paths relative to /root/reference).  The op ORDER of the floating-point
arithmetic follows the reference so results agree to the last few ulps.
"""
import math

import torch

_C = -0.5 * math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------
# zhusuan/distributions/normal.py
# ----------------------------------------------------------------------------
def normal_sample(mean, std, eps, n_samples=None, is_reparameterized=True):
    """Normal._sample, zhusuan/distributions/normal.py:89-107.

    ``eps`` replaces the reference's torch.normal draw and has shape
    ``[n_samples] + mean.shape`` (just ``mean.shape`` for n_samples None/1): the
    draw has MEAN's shape, not the broadcast shape (normal.py:91-92,104).
    The non-reparameterised branch (:102) is ``torch.normal(mean, std)``, equal
    to ``mean + std * randn`` bit-for-bit, detached from (mean, std).
    """
    k = 1 if n_samples is None else int(n_samples)
    if k > 1:
        reps = [k] + [1] * mean.dim()
        m = mean.repeat(reps)
        s = std.repeat(reps)
    else:
        m, s = mean, std
    z = m + s * eps
    if not is_reparameterized:
        z = z.detach()
    return z


def normal_log_prob(mean, std, x, group_ndims=0):
    """Normal._log_prob (normal.py:109-126) followed by the group sum of
    Distribution.log_prob (zhusuan/distributions/base.py:175-176)."""
    if x.dim() > mean.dim():  # normal.py:112-116: repeat params along the leading sample axis
        reps = [x.shape[0]] + [1] * mean.dim()
        m = mean.repeat(reps)
        s = std.repeat(reps)
    else:
        m, s = mean, std
    logstd = torch.log(s)
    c = torch.tensor(_C, dtype=x.dtype)
    precision = torch.exp(-2 * logstd)
    lp = c - logstd - 0.5 * precision * ((x - m) ** 2)
    if group_ndims > 0:
        lp = torch.sum(lp, [i for i in range(-group_ndims, 0)])
    return lp


# ----------------------------------------------------------------------------
# zhusuan/distributions/bernoulli.py
# ----------------------------------------------------------------------------
def bernoulli_probs_from_logits(logits):
    """Bernoulli.__init__ logits branch, bernoulli.py:46-50."""
    return torch.sigmoid(logits)


def bernoulli_logits_from_probs(probs):
    """Bernoulli.__init__ probs branch, bernoulli.py:43-45."""
    return torch.log(probs / (torch.ones(probs.shape) - probs))


def bernoulli_log_prob(probs, x, group_ndims=0):
    """Bernoulli._log_prob (bernoulli.py:84-95) + group sum (base.py:175-176).
    The ``+ 1e-8`` terms are part of the contract."""
    if x.dim() > probs.dim():  # bernoulli.py:88-90
        p = probs * torch.ones((x.shape[0],) + tuple(probs.shape))
    else:
        p = probs
    lp = x * torch.log(p + 1e-8) + (1 - x) * torch.log(1 - p + 1e-8)
    if group_ndims > 0:
        lp = torch.sum(lp, [i for i in range(-group_ndims, 0)])
    return lp


# ----------------------------------------------------------------------------
# zhusuan/distributions/logistic.py, uniform.py   
# ----------------------------------------------------------------------------
def _repeat_leading(k, *params):
    """The reference's ``p.repeat([n_samples, *len(first.shape) * [1]])`` (logistic.py:56-58, uniform.py:55-57):
    every parameter is repeated with the FIRST parameter's number of axes."""
    nd = params[0].dim()
    return [p.repeat([k] + [1] * nd) for p in params]


def logistic_sample(loc, scale, u, n_samples=None):
    """Logistic._sample, logistic.py:52-67.  ``u`` replaces the uniform draw (:64) and has shape
    ``[n_samples] + loc.shape`` (LOC's shape, :54-55,61)."""
    k = 1 if n_samples is None else int(n_samples)
    if k > 1:
        lo, sc = _repeat_leading(k, loc, scale)
    else:
        lo, sc = loc, scale
    epsilon = torch.log(u) - torch.log(1 - u)
    return lo + sc * epsilon


def logistic_log_prob(loc, scale, x, group_ndims=0):
    """Logistic._log_prob (logistic.py:69-83) + group sum (base.py:175-176)."""
    if x.dim() > loc.dim():
        lo, sc = _repeat_leading(x.shape[0], loc, scale)
    else:
        lo, sc = loc, scale
    z = (x - lo) / sc
    lp = -z - 2. * torch.nn.Softplus()(-z) - torch.log(sc)
    if group_ndims > 0:
        lp = torch.sum(lp, [i for i in range(-group_ndims, 0)])
    return lp


def uniform_sample(low, high, u, n_samples=None, is_reparameterized=True):
    """Uniform._sample, uniform.py:51-70.  Returns (sample, sample_cache).  ``u`` replaces the U(0,1) draw
    behind torch.distributions.Uniform.sample(): reparameterised -> shape ``[n_samples] + low.shape`` (:66-67),
    the cache holds u itself (:69); otherwise the draw has the broadcast shape of (low, high), is scaled into
    [low, high) (:64), cached, and scaled AGAIN by the return statement (:70)."""
    k = 1 if n_samples is None else int(n_samples)
    if k > 1:
        lo, hi = _repeat_leading(k, low, high)
    else:
        lo, hi = low, high
    if not is_reparameterized:
        with torch.no_grad():
            cache = lo + u * (hi - lo)
    else:
        cache = u
    return cache * (hi - lo) + lo, cache


def uniform_log_prob(low, high, x, group_ndims=0):
    """Uniform._log_prob (uniform.py:72-85): torch.distributions.Uniform(low, high).log_prob(x) =
    log(lb * ub) - log(high - low) with lb = (low <= x), ub = (high > x); + group sum (base.py:175-176)."""
    if x.dim() > low.dim():
        lo, hi = _repeat_leading(x.shape[0], low, high)
    else:
        lo, hi = low, high
    lb = lo.le(x).type_as(lo)
    ub = hi.gt(x).type_as(lo)
    lp = torch.log(lb.mul(ub)) - torch.log(hi - lo)
    if group_ndims > 0:
        lp = torch.sum(lp, [i for i in range(-group_ndims, 0)])
    return lp


# ----------------------------------------------------------------------------
# zhusuan/framework/stochastic_tensor.py
# ----------------------------------------------------------------------------
def st_reduce(lp, reduce_mean_dims=None, reduce_sum_dims=None, multiplier=None):
    """StochasticTensor.log_prob post-processing, stochastic_tensor.py:160-181:
    mean dims (keepdim) -> sum dims (keepdim) -> squeeze those dims in
    descending order -> multiplier.  (The ``shape == [1]`` early exit at :174
    never fires: torch.Size([1]) == [1] is False.)"""
    if reduce_mean_dims:
        lp = torch.mean(lp, reduce_mean_dims, keepdim=True)
    if reduce_sum_dims:
        lp = torch.sum(lp, reduce_sum_dims, keepdim=True)
    if reduce_mean_dims or reduce_sum_dims:
        dims = list(reduce_mean_dims or []) + list(reduce_sum_dims or [])
        dims.sort(reverse=True)
        for d in dims:
            lp = torch.squeeze(lp, d)
    if multiplier:
        lp = lp * multiplier
    return lp


# ----------------------------------------------------------------------------
# zhusuan/utils.py
# ----------------------------------------------------------------------------
def log_mean_exp(x, dim=None, keepdims=False):
    """zhusuan/utils.py:6-21."""
    x_max = torch.max(x, dim, True).values
    ret = torch.log(torch.mean(torch.exp(x - x_max), dim, True)) + x_max
    if not keepdims:
        ret = torch.mean(ret, dim=dim)
    return ret


# ----------------------------------------------------------------------------
# zhusuan/variational/elbo.py
# ----------------------------------------------------------------------------
def elbo_sgvb(logpxz, logqz, reduce_mean=True):
    """ELBO.sgvb without a flow transform, elbo.py:155-161."""
    if logqz.dim() > 0 and reduce_mean:
        elbo = torch.mean(logpxz - logqz)
    else:
        elbo = logpxz - logqz
    return -elbo


def elbo_reinforce(logpxz, logqz, moving_mean, local_step, reduce_mean=True, baseline=None, variance_reduction=True,
                   decay=0.8):
    """ELBO.reinforce, elbo.py:163-238.  ``moving_mean`` (float32 [1]) and ``local_step`` (int32 [1]) are the module
    buffers of elbo.py:45-49 and are updated IN PLACE, including the division of the moving mean by the bias factor
    on every call (:224).  Returns the cost, or (loss, mean elbo) when a baseline is used."""
    decay_tensor = torch.ones(size=[1], dtype=torch.float32) * decay
    l_signal = (logpxz - logqz).detach()
    baseline_cost = None
    vector_mean = len(logqz.shape) > 0 and reduce_mean
    if variance_reduction:
        if baseline is not None:
            baseline_cost = 0.5 * torch.square(l_signal.detach() - baseline)
            if vector_mean:
                baseline_cost = torch.mean(baseline_cost)
            l_signal = l_signal - baseline
        bc = torch.mean(l_signal) if vector_mean else l_signal
        moving_mean -= (moving_mean - bc.detach()) * (1.0 - decay)
        local_step += 1
        bias_factor = 1 - torch.pow(decay_tensor, local_step)
        moving_mean /= bias_factor
        l_signal = l_signal.detach().clone()
        l_signal -= moving_mean.detach()      # in place as in the reference: a 0-d l_signal raises here (:225)
    l_signal = l_signal.detach()
    cost = -(logpxz + l_signal * logqz)
    if baseline_cost is not None:
        loss = torch.mean(cost + baseline_cost) if vector_mean else cost + baseline_cost
        return loss, torch.mean(logpxz - logqz)
    return torch.mean(cost) if vector_mean else cost


# ----------------------------------------------------------------------------
# zhusuan/variational/importance_weighted_objective.py
# ----------------------------------------------------------------------------
def iw_term(log_w, axis):
    """compute_iw_term, importance_weighted_objective.py:16-25."""
    m = torch.max(log_w, axis, keepdim=True).values
    w = (log_w - m).exp()
    w_tilde = (w / w.sum(dim=axis, keepdim=True)).detach()
    return (w_tilde * log_w).sum(axis)


def iw_sgvb(logpxz, logqz, axis=0, reduce_mean=True):
    """ImportanceWeightedObjective.sgvb, importance_weighted_objective.py:123-132."""
    lower = iw_term(logpxz - logqz, axis)
    return torch.mean(-lower) if reduce_mean else -lower


def iw_vimco(logpxz, logqz, axis=0):
    """ImportanceWeightedObjective.vimco, importance_weighted_objective.py:152-191,
    for 1-D or 2-D log_w with the particle axis first (the only layouts the
    reference's permutation code handles).

    For column j the control variate is log_mean_exp over the K values
    {l_i : i != j} U {mean_{i != j} l_i}  (the reference builds this as a
    [B, K, K] tensor with the diagonal replaced, :184-186)."""
    if axis != 0 or logpxz.dim() > 2:
        raise ValueError("oracle.iw_vimco handles axis=0, 1-D / 2-D inputs")
    log_w = logpxz - logqz
    k = log_w.shape[0]
    if k < 2:
        raise ValueError("VIMCO needs at least two particles")
    one_d = log_w.dim() == 1
    l = log_w.reshape(k, -1)                                  # [K, B]
    sub = (torch.sum(l, dim=0, keepdim=True) - l) / torch.as_tensor(k - 1, dtype=l.dtype)
    x = l.t()                                                 # [B, K]  (i index)
    x_ex = x.unsqueeze(2).repeat(1, 1, k)                     # x_ex[b, i, j] = l[i, b]
    x_ex = x_ex - torch.diag_embed(x) + torch.diag_embed(sub.t())
    cv = log_mean_exp(x_ex, 1).t()                            # [K, B], index j
    signal = log_mean_exp(l, 0, keepdims=True) - cv
    if one_d:
        signal = signal.reshape(k)
    fake = torch.sum(logqz * signal.detach(), 0)
    cost = -fake - iw_term(log_w, 0)
    return cost.mean()


# ----------------------------------------------------------------------------
# Callers (examples/*) as parameter-dict functions.  ``eps`` is the list of
# Gaussian draws in the order the reference consumes them: every latent is
# drawn when its node is created and again when the objective re-reads
# ``node.tensor`` (elbo.py:122, importance_weighted_objective.py:85); the
# second draw is the one used.
# ----------------------------------------------------------------------------
def _mlp(x, ws, bs, acts):
    h = x
    for w, b, a in zip(ws, bs, acts):
        h = torch.nn.functional.linear(h, w, b)
        if a == "relu":
            h = torch.relu(h)
        elif a == "sigmoid":
            h = torch.sigmoid(h)
    return h


def vae_loss(p, x, eps_used):
    """examples/variational_autoencoder/vae_mnist.py:31-86 under ELBO.forward
    (elbo.py:81-132).  ``p`` maps the reference's parameter names to tensors."""
    B = x.shape[0]
    h = _mlp(x, [p["variational.sq.0.weight"], p["variational.sq.2.weight"]],
             [p["variational.sq.0.bias"], p["variational.sq.2.bias"]], ["relu", "relu"])
    z_mean = torch.nn.functional.linear(h, p["variational.fc3.weight"], p["variational.fc3.bias"])
    z_std = torch.exp(torch.nn.functional.linear(h, p["variational.fc4.weight"], p["variational.fc4.bias"]))
    z = normal_sample(z_mean, z_std, eps_used)
    logqz = st_reduce(normal_log_prob(z_mean, z_std, z), [0], [1])
    zd = z.shape[1]
    logpz = st_reduce(normal_log_prob(torch.zeros([B, zd]), torch.ones([B, zd]), z), [0], [1])
    x_probs = _mlp(z, [p["generator.sequential.0.weight"], p["generator.sequential.2.weight"],
                       p["generator.sequential.4.weight"]],
                   [p["generator.sequential.0.bias"], p["generator.sequential.2.bias"],
                    p["generator.sequential.4.bias"]], ["relu", "relu", "sigmoid"])
    logpx = st_reduce(bernoulli_log_prob(x_probs, x), [0], [1])
    loss = elbo_sgvb(logpz + logpx, logqz)
    return loss, dict(z=z, logqz=logqz, logpz=logpz, logpx=logpx, x_mean=x_probs)


def iwae_loss(p, x, eps_used, n_samples, estimator):
    """examples/variational_autoencoder/iwae.py:49-120 under
    ImportanceWeightedObjective.forward (importance_weighted_objective.py:79-100)."""
    B = x.shape[0]
    h = _mlp(x, [p["variational.output_logits.0.weight"], p["variational.output_logits.2.weight"]],
             [p["variational.output_logits.0.bias"], p["variational.output_logits.2.bias"]], ["relu", "relu"])
    z_mean = torch.nn.functional.linear(h, p["variational.output_mean.weight"], p["variational.output_mean.bias"])
    z_std = torch.exp(torch.nn.functional.linear(h, p["variational.output_logstd.weight"],
                                                 p["variational.output_logstd.bias"]))
    reparam = estimator == "sgvb"
    z = normal_sample(z_mean, z_std, eps_used, n_samples, reparam)
    logqz = st_reduce(normal_log_prob(z_mean, z_std, z), None, [2])
    zd = z.shape[-1]
    logpz = st_reduce(normal_log_prob(torch.zeros([B, zd]), torch.ones([B, zd]), z), None, [2])
    x_probs = _mlp(z, [p["generator.gen_sq.0.weight"], p["generator.gen_sq.2.weight"], p["generator.gen_sq.4.weight"]],
                   [p["generator.gen_sq.0.bias"], p["generator.gen_sq.2.bias"], p["generator.gen_sq.4.bias"]],
                   ["relu", "relu", "sigmoid"])
    logpx = st_reduce(bernoulli_log_prob(x_probs, x), None, [2])
    logpxz = logpz + logpx
    if estimator == "sgvb":
        loss = iw_sgvb(logpxz, logqz, 0)
    else:
        loss = iw_vimco(logpxz, logqz, 0)
    log_w = (logpxz - logqz).detach()
    return loss, dict(z=z, logqz=logqz, logpz=logpz, logpx=logpx, log_w=log_w,
                      iw_bound=log_mean_exp(log_w, 0).mean())


def bnn_loss(w_means, w_logstds, y_logstd, x, y, eps_used, n_particles, multiplier=456):
    """examples/bayesian_neural_nets/bnn_vi.py:24-99 under ELBO.forward.
    ``eps_used`` = second draws, one per layer, shape [K, n_out, n_in + 1]."""
    K = n_particles
    B = x.shape[0]
    logq = None
    logp = None
    h = x.repeat([K] + [1] * x.dim())
    n_layers = len(w_means)
    for i in range(n_layers):
        sd = torch.exp(w_logstds[i])
        w = normal_sample(w_means[i], sd, eps_used[i], K)
        lq = st_reduce(normal_log_prob(w_means[i], sd, w, 2), [0])
        logq = lq if logq is None else logq + lq
        shp = tuple(w_means[i].shape)
        lpw = st_reduce(normal_log_prob(torch.zeros(shp), torch.ones(shp), w, 2), [0])
        logp = lpw if logp is None else logp + lpw
        wr = torch.unsqueeze(w, 1).repeat([1, B, 1, 1])
        h = torch.cat((h, torch.ones([*h.shape[:-1], 1])), -1)
        h = torch.unsqueeze(h, -1)
        scale = torch.sqrt(torch.as_tensor(h.shape[2], dtype=torch.float32))
        h = torch.matmul(wr, h) / scale
        h = torch.squeeze(h, -1)
        if i < n_layers - 1:
            h = torch.relu(h)
    y_mean = torch.squeeze(h, 2)
    rmse = torch.sqrt(torch.mean((y - torch.mean(y_mean, 0)) ** 2))
    lpy = st_reduce(normal_log_prob(y_mean, torch.exp(y_logstd), y), [0, 1], None, multiplier)
    logp = logp + lpy
    return elbo_sgvb(logp, logq), dict(rmse=rmse, logp_y=lpy)
