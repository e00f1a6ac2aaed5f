"""Evidence lower bound objective.  Interface of zhusuan/variational/elbo.py:5-253 of the reference
(``ELBO(generator, variational, estimator='sgvb', ...)``, ``model(observed, reduce_mean=True)``)."""
import torch
import torch.nn as nn

from .. import _hip, _ops, _rng
from ..distributions.normal import Normal
from ..framework import stochastic_tensor as _st

__all__ = ['ELBO', 'EvidenceLowerBoundObjective']

# the one-launch sampler (MS1) is for the launch-bound shapes: above this many elements in total each node takes K1
_MULTI_DRAW_MAX_ELEMENTS = 1 << 20
# ... and so is the element-wise part of the one-launch objective (LJ1): beyond this many elements the nodes' own streaming
# kernels (K2 / K3: 72-83 % of the HBM roofline at that size) evaluate the log-probs and LJ1 only adds their rows up
_LOGJOINT_MAX_ELEMENTS = 1 << 22


def latent_value(node):
    """``node.tensor`` as handed to the generator.  For Normal / Bernoulli a non-reparameterised draw carries a grad_fn
    whose derivative is identically zero (``torch.normal(mean, std)``, normal.py:102); detaching it here changes no
    gradient and lets autograd skip the dead branches through the generator (first decoder dgrad, prior log-prob
    backward).  Families whose non-reparameterised draw still depends differentiably on the parameters (Uniform,
    uniform.py:63-70) stay attached, so the pathwise gradient reaches low / high as in the reference."""
    _st.expire_deferred_value(node)
    t = node.tensor
    dist = getattr(node, 'dist', None)
    if (dist is not None and not dist.is_reparameterized and isinstance(t, torch.Tensor)
            and getattr(dist, '_nonreparam_draw_has_zero_grad', False)):
        return t.detach()
    return t


def run_variational(net, observed):
    """``net(observed)`` as the objectives call it (elbo.py:88).  Inside ``zhusuan.skip_discarded_draws()`` the node
    factories of the net do not sample while it runs (their first draw is the one every objective throws away,
    elbo.py:122): they hand out ``LazyDraw`` stand-ins that sample only if the net's own code touches them."""
    if _st.skipping_discarded_draws():
        with _st.deferred_node_values():
            return net(observed)
    # default: both draws of every latent are executed, as in the reference -- where the sampling kernel takes the shape, in ONE
    # launch at the node's creation (distributions/normal.py: the second draw waits for the re-read)
    with _rng.expecting_redraw():
        return net(observed)


def draw_latents(nodes_q):
    """``{name: node.tensor}`` for the latents of the variational net -- the objectives' re-read of every node
    (elbo.py:122, importance_weighted_objective.py:85): a FRESH draw per node, in node order.  Two or more unobserved,
    reparameterised Normal nodes of launch-bound size are drawn by ONE launch (MS1, ``NormalSampleLogProbMulti``); every
    other node draws by itself (``latent_value``).  Draw order, injected epsilons and Philox call ids are those of the
    node-by-node loop."""
    names = list(nodes_q.keys())
    if any(getattr(getattr(v, 'dist', None), '__dict__', {}).get('_pending_draw') is not None for v in nodes_q.values()):
        return {k: latent_value(v) for k, v in nodes_q.items()}        # (second draws made with the first: nothing to batch)
    batch = []
    for name in names:
        node = nodes_q[name]
        dist = getattr(node, 'dist', None)
        if (type(dist) is Normal and dist.is_reparameterized and name not in node.bn.observed
                and tuple(dist._mean.shape) == tuple(dist._scale_operand().shape) and dist._mean.numel() > 0):
            batch.append(name)
    total = sum(max(int(nodes_q[n]._n_samples or 1), 1) * nodes_q[n].dist._mean.numel() for n in batch)
    devs = {(nodes_q[n].dist._mean.device, nodes_q[n].dist._mean.dtype) for n in batch}
    if len(batch) < 2 or len(batch) > _hip.MS_MAX_TERMS or total > _MULTI_DRAW_MAX_ELEMENTS or len(devs) != 1 \
            or _rng.reference_stream_active():
        return {k: latent_value(v) for k, v in nodes_q.items()}
    values = {}
    # nodes before the first batched one keep their place in the draw order; the batched ones are consecutive from there on
    # only if nothing else draws in between -- otherwise fall back to the plain loop
    idx = [names.index(n) for n in batch]
    if idx != list(range(idx[0], idx[0] + len(batch))):
        return {k: latent_value(v) for k, v in nodes_q.items()}
    for name in names[:idx[0]]:
        values[name] = latent_value(nodes_q[name])
    meta, tensors, plans = [], [], []
    seed, rng_state = 0, None
    for name in batch:
        node = nodes_q[name]
        _st.expire_deferred_value(node)
        mu, sigma, eps, K, has_k, n_fold, is_logstd, _ = node.dist._sample_plan(1 if node._n_samples is None else node._n_samples)
        call = 0
        if eps is None:
            seed, call, rng_state = _rng.next_call(mu.device)
        meta.append((K if has_k else 1, has_k, n_fold, is_logstd, call))
        tensors += [mu, sigma, eps]
        plans.append((node, n_fold))
    outs = _ops.NormalSampleLogProbMulti.apply(tuple(meta), seed, rng_state, *tensors)
    for i, (node, n_fold) in enumerate(plans):
        z, lp = outs[2 * i], outs[2 * i + 1]
        z._zs_grad_alias = outs[2 * len(plans) + i]      # (the same sample as a second output: see NormalSampleLogProbMulti)
        node.dist._adopt_draw(z, lp, n_fold)
        values[node.name] = z
    for name in names[idx[-1] + 1:]:
        values[name] = latent_value(nodes_q[name])
    return {k: values[k] for k in names}


from ..utils import note_path as _note_path      # noqa: E402

_PATH_NODES = "per-node kernels: one log-probability launch per node (+ torch's reductions), each way"


class ELBO(nn.Module):
    """
    :param generator: BayesianNet p(x, z).
    :param variational: BayesianNet q(z | x).
    :param estimator: 'sgvb' (reparameterisation) or 'reinforce' (score function).
    :param transform: normalising-flow transform of the latents -- outside the hot path of this
        build (elbo.py:90-119 depends on zhusuan.invertible): NotImplementedError.
    """

    def __init__(self, generator, variational, estimator='sgvb', transform=None, transform_var=[],
                 auxillary_var=[]):
        super(ELBO, self).__init__()
        self.generator = generator
        self.variational = variational
        if estimator not in ['sgvb', 'reinforce']:
            raise NotImplementedError()
        self.estimator = estimator
        if estimator == 'reinforce':
            self.register_buffer('moving_mean', torch.zeros(size=[1], dtype=torch.float32))
            self.register_buffer('local_step', torch.zeros(size=[1], dtype=torch.int32))
        if transform is not None:
            raise NotImplementedError(
                "ELBO(transform=...) relies on zhusuan.invertible flows, which are outside the "
                "variational-inference hot path of the MI355X build")
        self.transform = None
        self.last_path = None          # which kernels the last evaluation ran on, and why (zhusuan.explain)

    def log_joint(self, nodes):
        """Sum of node log-probs in insertion order (elbo.py:58-79)."""
        log_joint_ = None
        for n_name in nodes.keys():
            lp = nodes[n_name].log_prob()
            log_joint_ = lp if log_joint_ is None else log_joint_ + lp
        return log_joint_

    def forward(self, observed, reduce_mean=True, **kwargs):
        """elbo.py:81-132: run q; re-read every latent's ``.tensor`` (a second, fresh draw -- the one
        that is used); run p on {latents} U observed; combine the two log-joints."""
        run_variational(self.variational, observed)
        nodes_q = self.variational.nodes
        _v_inputs = draw_latents(nodes_q)
        _observed = {**_v_inputs, **observed}
        self.generator(_observed)
        nodes_p = self.generator.nodes
        if self.estimator == "sgvb" and type(self).sgvb is ELBO.sgvb and type(self).log_joint is ELBO.log_joint:
            why = []
            fused = self._scalar_sgvb(nodes_p, nodes_q, why)      # (a subclass that overrides either hook keeps its hooks)
            if fused is not None:
                _note_path(self, "LJ1: every node's log-probability, the reductions and the objective in one launch each way (zs_logjoint_scalar)")
                return fused
            _note_path(self, _PATH_NODES, why[0] if why else "some node's log-probability is not a scalar")
        else:
            _note_path(self, _PATH_NODES, "the 'reinforce' estimator takes its own one-launch epilogue (R1) after the per-node kernels"
                       if self.estimator != "sgvb" else "a subclass overrides log_joint / sgvb: its hooks are called as the reference calls them")
        logpxz = self.log_joint(nodes_p)
        logqz = self.log_joint(nodes_q)
        if self.estimator == "sgvb":
            return self.sgvb(logpxz, logqz, reduce_mean)
        return self.reinforce(logpxz, logqz, reduce_mean, **kwargs)

    def _scalar_sgvb(self, nodes_p, nodes_q, why=None):
        """When every node's log-probability reduces to a scalar (the VAE and BNN callers), the whole sgvb objective
        -(sum_p log p - sum_q log q) is ONE launch (LJ1, ``zs_logjoint_scalar``): the element-wise log-probs of every
        Normal / Bernoulli node, the fused log-densities the sampling kernel has already produced, their reductions
        (mean / sum over the reduce dims, multiplier: one coefficient per node) and the scalar arithmetic of
        elbo.py:58-79,155-161 -- instead of one log-prob launch per node plus a mean / sum / multiply per node and the
        adds; the backward of all of it is one more launch.  Returns None when some node keeps a non-scalar shape."""
        why = [] if why is None else why
        plan = [(sign, nodes[name]) for sign, nodes in ((-1.0, nodes_p), (1.0, nodes_q)) for name in nodes.keys()]
        if not plan or len(plan) > _hip.LJ_MAX_TERMS:
            why.append("the two nets have %d nodes (the one-launch log-joint takes 1 .. %d)" % (len(plan), _hip.LJ_MAX_TERMS))
            return None
        for _, node in plan:                        # decide first (no kernel is launched by the question)
            if not hasattr(node, '_scalar_coef') or node._scalar_coef() is None:
                why.append("node %r keeps a non-scalar log-probability (or is of a family without a one-launch term)"
                           % getattr(node, 'name', '?'))
                return None
        def collect(rows_only):
            spec, tensors = [], []
            for sign, node in plan:
                fam, operands, periods, n, coef = node._scalar_term(rows_only=rows_only)
                spec.append((fam, sign * coef, n) + tuple(periods))
                tensors += list(operands)
            return spec, tensors
        spec, tensors = collect(False)
        if sum(t[2] for t in spec if t[0] != _hip.LJ_ROWS) > _LOGJOINT_MAX_ELEMENTS:
            spec, tensors = collect(True)
        if len({t.dtype for t in tensors if t is not None}) != 1 or len({t.device for t in tensors if t is not None}) != 1:
            why.append("the nodes differ in dtype or device")
            return None
        return _ops.LogJointScalar.apply(tuple(spec), *tensors)

    def sgvb(self, logpxz, logqz, reduce_mean=True, log_det=None):
        """elbo.py:134-161."""
        if len(logqz.shape) > 0 and reduce_mean:
            elbo = torch.mean(logpxz - logqz)
        else:
            elbo = logpxz - logqz
        if log_det is not None:
            elbo = elbo + torch.mean(torch.sum(log_det)).squeeze()
        return -elbo

    def reinforce(self, logpxz, logqz, reduce_mean=True, baseline=None, variance_reduction=True, decay=0.8):
        """Score-function estimator with moving-mean baseline (elbo.py:163-238) as ONE kernel (R1,
        ``zs_reinforce_f32``): learning signal, moving-mean update -- including the reference's in-place division of
        ``moving_mean`` by the bias factor on every call (:224) -- cost and its mean.  The moving mean and the step
        counter stay on the device, so the objective can be captured in a hipGraph."""
        logpxz = torch.as_tensor(logpxz)
        logqz = torch.as_tensor(logqz, device=logpxz.device)
        if logpxz.shape != logqz.shape:
            logpxz, logqz = torch.broadcast_tensors(logpxz, logqz)
        vector = logqz.dim() > 0
        do_mean = vector and bool(reduce_mean)
        if variance_reduction:
            # shapes the reference's in-place buffer arithmetic cannot broadcast (elbo.py:221,225)
            if not vector:
                raise RuntimeError("output with shape [] doesn't match the broadcast shape [1]")
            if not do_mean and logqz.numel() != 1:
                raise RuntimeError("output with shape [1] doesn't match the broadcast shape %s" % list(logqz.shape))
        use_b = bool(variance_reduction) and baseline is not None
        if use_b:
            baseline = torch.as_tensor(baseline, dtype=logqz.dtype, device=logqz.device)
        cost = _ops.ReinforceEpilogue.apply(logpxz, logqz, baseline if use_b else None, self.moving_mean, self.local_step,
                                            bool(variance_reduction), do_mean, float(decay))
        if use_b:
            return cost, torch.mean(logpxz - logqz)
        return cost


class EvidenceLowerBoundObjective(ELBO):
    """Alias of ELBO (elbo.py:241-253)."""
    pass
