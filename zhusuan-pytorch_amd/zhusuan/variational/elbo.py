"""Evidence lower bound objective.  Interface of zhusuan/variational/elbo.py:5-253 of the reference
(``ELBO(generator, variational, estimator='sgvb', ...)``, ``model(observed, reduce_mean=True)``)."""
import torch
import torch.nn as nn

__all__ = ['ELBO', 'EvidenceLowerBoundObjective']


def latent_value(node):
    """``node.tensor`` as handed to the generator.  A non-reparameterised draw carries a grad_fn whose
    derivative is identically zero (``torch.normal(mean, std)``, normal.py:102); detaching it here changes no
    gradient and lets autograd skip the dead branches through the generator (first decoder dgrad, prior
    log-prob backward)."""
    t = node.tensor
    dist = getattr(node, 'dist', None)
    if dist is not None and not dist.is_reparameterized and isinstance(t, torch.Tensor):
        return t.detach()
    return t


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
        self.variational(observed)
        nodes_q = self.variational.nodes
        _v_inputs = {k: latent_value(v) for k, v in nodes_q.items()}
        _observed = {**_v_inputs, **observed}
        self.generator(_observed)
        nodes_p = self.generator.nodes
        logpxz = self.log_joint(nodes_p)
        logqz = self.log_joint(nodes_q)
        if self.estimator == "sgvb":
            return self.sgvb(logpxz, logqz, reduce_mean)
        return self.reinforce(logpxz, logqz, reduce_mean, **kwargs)

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
        """Score-function estimator with moving-mean baseline (elbo.py:163-238), including the
        reference's in-place division of ``moving_mean`` by the bias factor each step (:224)."""
        dev = logqz.device
        decay_tensor = torch.ones(size=[1], dtype=torch.float32, device=dev) * decay
        l_signal = (logpxz - logqz).detach()
        baseline_cost = None
        if variance_reduction:
            if baseline is not None:
                baseline_cost = 0.5 * torch.square(l_signal.detach() - baseline)
                if len(logqz.shape) > 0 and reduce_mean:
                    baseline_cost = torch.mean(baseline_cost)
                l_signal = l_signal - baseline
            if len(logqz.shape) > 0 and reduce_mean:
                bc = torch.mean(l_signal)
            else:
                bc = l_signal
            self.moving_mean -= (self.moving_mean - bc.detach()) * (1.0 - decay)
            self.local_step += 1
            bias_factor = 1 - torch.pow(decay_tensor, self.local_step)
            self.moving_mean /= bias_factor
            l_signal = l_signal - self.moving_mean.detach()
        l_signal = l_signal.detach()
        cost = -(logpxz + l_signal * logqz)
        if baseline_cost is not None:
            if len(logqz.shape) > 0 and reduce_mean:
                loss = torch.mean(cost + baseline_cost)
            else:
                loss = cost + baseline_cost
            return loss, torch.mean(logpxz - logqz)
        if len(logqz.shape) > 0 and reduce_mean:
            cost = torch.mean(cost)
        return cost


class EvidenceLowerBoundObjective(ELBO):
    """Alias of ELBO (elbo.py:241-253)."""
    pass
