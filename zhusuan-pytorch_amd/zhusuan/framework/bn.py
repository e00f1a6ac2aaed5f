"""BayesianNet: an nn.Module holding named stochastic nodes, a cache of deterministic values and the
observations.  Interface of zhusuan/framework/bn.py:22-240 of the reference."""
import torch
import torch.nn as nn

from .stochastic_tensor import StochasticTensor, node_value
from ..distributions import (Distribution, Normal, Bernoulli, Logistic, Uniform, Beta, Exponential, Gamma, Laplace, Poisson,
                             StudentT)

__all__ = ['BayesianNet']

# bn.py:8-19: the ten names (Normal / Bernoulli / Logistic / Uniform run on the HIP kernels, the other six are
# torch.distributions pass-throughs off the hot path)
name_mapping = {
    "Normal": Normal,
    "Bernoulli": Bernoulli,
    "Beta": Beta,
    "Exponential": Exponential,
    "Gamma": Gamma,
    "Laplace": Laplace,
    "Logistic": Logistic,
    "Poisson": Poisson,
    "StudentT": StudentT,
    "Uniform": Uniform,
}


class BayesianNet(nn.Module):
    """
    Build a model by subclassing and creating nodes inside ``forward``::

        class Net(BayesianNet):
            def forward(self, observed):
                self.observe(observed)
                z = self.normal('z', mean=..., std=..., n_samples=K, reduce_sum_dims=[2])
                ...
                return self

    :param observed: dict name -> tensor of observed node values.
    """

    def __init__(self, observed=None, device=torch.device('cpu')):
        super(BayesianNet, self).__init__()
        self._nodes = {}
        self._cache = {}
        self._observed = observed if observed else {}
        self._device = device

    @property
    def nodes(self):
        return self._nodes

    @property
    def cache(self):
        return self._cache

    @property
    def observed(self):
        return self._observed

    @property
    def device(self):
        """Device of the first parameter, else the one given to the constructor / ``to`` (bn.py:91-101).
        Cached: walking ``parameters()`` costs ~25 us and forward() asks several times per step."""
        d = self.__dict__.get('_device_cache')
        if d is None:
            try:
                d = next(self.parameters()).device
            except StopIteration:
                d = self._device
            self.__dict__['_device_cache'] = d
        return d

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_device_cache'] = None       # .to() / .cuda() / .cpu() of this module or a parent
        return super()._apply(fn, *args, **kwargs)

    def to(self, device):
        self._device = torch.device(device) if not isinstance(device, torch.device) else device
        self.__dict__['_device_cache'] = None
        return super().to(device)

    def observe(self, observed):
        """Replace the observation dict (bn.py:115-125)."""
        self._observed = {}
        for k, v in observed.items():
            self._observed[k] = v
        return self

    def sn(self, dist, name, n_samples=None, **kwargs):
        return self.stochastic_node(dist, name, n_samples, **kwargs)

    def snode(self, *args, **kwargs):
        return self.stochastic_node(*args, **kwargs)

    def stochastic_node(self, distribution, name, n_samples=None, **kwargs):
        """Add (or overwrite) node `name`; returns its current value (observation or fresh sample),
        bn.py:139-158."""
        if isinstance(distribution, str):
            _dist = name_mapping[distribution](device=self.device, **kwargs)
            self._nodes[name] = StochasticTensor(self, name, _dist, n_samples=n_samples, **kwargs)
        elif isinstance(distribution, Distribution):
            distribution._device = self.device
            self._nodes[name] = StochasticTensor(self, name, distribution, n_samples=n_samples, **kwargs)
        else:
            raise ValueError('distribution must be name of sub class of Distribution or an instance of Distribution')
        return node_value(self._nodes[name])

    def _log_joint(self):
        ret = 0
        for k, v in self._nodes.items():
            if isinstance(v, StochasticTensor):
                ret = ret + v.log_prob()
        return ret

    def log_joint(self, use_cache=False):
        """Sum of the log-probs of all nodes at their current values (bn.py:170-185)."""
        if use_cache:
            if not hasattr(self, '_log_joint_cache'):
                self._log_joint_cache = self._log_joint()
        else:
            self._log_joint_cache = self._log_joint()
        return self._log_joint_cache

    def normal(self, name, mean=0., std=None, logstd=None, dtype=None, is_continuous=True,
               is_reparameterized=True, group_ndims=0, n_samples=None, **kwargs):
        if not isinstance(name, str):
            raise ValueError("name of stochastic_node must be str")
        distribution = Normal(mean=mean, std=std, logstd=logstd, dtype=dtype, is_continuous=is_continuous,
                              is_reparameterized=is_reparameterized, group_ndims=group_ndims,
                              device=self.device, **kwargs)
        self._nodes[name] = StochasticTensor(self, name, distribution, n_samples=n_samples, **kwargs)
        return node_value(self._nodes[name])

    def bernoulli(self, name, logits=None, probs=None, dtype=None, is_continuous=False, group_ndims=0,
                  n_samples=None, **kwargs):
        if not isinstance(name, str):
            raise ValueError("name of stochastic_node must be str")
        distribution = Bernoulli(logits=logits, probs=probs, dtype=dtype, is_continuous=is_continuous,
                                 group_ndims=group_ndims, device=self.device, **kwargs)
        self._nodes[name] = StochasticTensor(self, name, distribution, n_samples=n_samples, **kwargs)
        return node_value(self._nodes[name])

    def uniform(self, name, low, high, dtype=None, is_continuous=True, is_reparameterized=True, group_ndims=0,
                n_samples=None, **kwargs):
        """bn.py:408-432."""
        if not isinstance(name, str):
            raise ValueError("name of stochastic_node must be str")
        distribution = Uniform(low=low, high=high, dtype=dtype, is_continuous=is_continuous,
                               is_reparameterized=is_reparameterized, group_ndims=group_ndims,
                               device=self.device, **kwargs)
        self._nodes[name] = StochasticTensor(self, name, distribution, n_samples=n_samples, **kwargs)
        return node_value(self._nodes[name])

    def logistic(self, name, loc, scale, dtype=None, is_continuous=True, group_ndims=0, n_samples=None, **kwargs):
        """As in the reference, the helper of this name builds a LAPLACE node (bn.py:336-358 constructs ``Laplace``).
        For a Logistic node use ``stochastic_node('Logistic', name, loc=..., scale=...)`` or pass a Logistic instance."""
        return self._family_node(Laplace, name, n_samples, dict(loc=loc, scale=scale, dtype=dtype, is_continuous=is_continuous,
                                                                group_ndims=group_ndims), kwargs)

    def _family_node(self, cls, name, n_samples, params, kwargs):
        if not isinstance(name, str):
            raise ValueError("name of stochastic_node must be str")
        distribution = cls(device=self.device, **params, **kwargs)
        self._nodes[name] = StochasticTensor(self, name, distribution, n_samples=n_samples, **kwargs)
        return node_value(self._nodes[name])

    # the torch.distributions pass-through families (bn.py:242-406): plain torch ops, off the hot path
    def beta(self, name, alpha, beta, dtype=None, is_continuous=True, group_ndims=0, n_samples=None, **kwargs):
        return self._family_node(Beta, name, n_samples, dict(alpha=alpha, beta=beta, dtype=dtype, is_continuous=is_continuous,
                                                             group_ndims=group_ndims), kwargs)

    def exponential(self, name, rate, dtype=None, is_continuous=True, group_ndims=0, n_samples=None, **kwargs):
        return self._family_node(Exponential, name, n_samples, dict(rate=rate, dtype=dtype, is_continuous=is_continuous,
                                                                    group_ndims=group_ndims), kwargs)

    def gamma(self, name, alpha, beta, dtype=None, is_continuous=True, group_ndims=0, n_samples=None, **kwargs):
        return self._family_node(Gamma, name, n_samples, dict(alpha=alpha, beta=beta, dtype=dtype, is_continuous=is_continuous,
                                                              group_ndims=group_ndims), kwargs)

    def laplace(self, name, loc, scale, dtype=None, is_continuous=True, group_ndims=0, n_samples=None, **kwargs):
        return self._family_node(Laplace, name, n_samples, dict(loc=loc, scale=scale, dtype=dtype, is_continuous=is_continuous,
                                                                group_ndims=group_ndims), kwargs)

    def poisson(self, name, rate, dtype=None, is_continuous=True, group_ndims=0, n_samples=None, **kwargs):
        return self._family_node(Poisson, name, n_samples, dict(rate=rate, dtype=dtype, is_continuous=is_continuous,
                                                                group_ndims=group_ndims), kwargs)

    def studentT(self, name, df, loc=0., scale=1., dtype=None, is_continuous=True, group_ndims=0, n_samples=None, **kwargs):
        return self._family_node(StudentT, name, n_samples, dict(df=df, loc=loc, scale=scale, dtype=dtype,
                                                                 is_continuous=is_continuous, group_ndims=group_ndims), kwargs)
