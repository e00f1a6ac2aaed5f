"""Names the reference's ``zhusuan.distributions`` also exports (zhusuan/distributions/__init__.py:6-13) but that are
outside this build: six thin wrappers of ``torch.distributions`` and ``FlowDistribution`` (needs ``zhusuan.invertible``).
They exist so that ``from zhusuan.distributions import *`` style imports of existing model code keep working; using one
raises with a pointer to what is built."""
from .base import Distribution

__all__ = ['Beta', 'Exponential', 'Gamma', 'Laplace', 'Poisson', 'StudentT', 'FlowDistribution']


def _placeholder(name):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "zhusuan.distributions.%s is outside the variational-inference hot path of the MI355X build "
            "(built: Normal, Bernoulli, Logistic, Uniform)" % name)
    return type(name, (Distribution,), {"__init__": __init__, "__doc__": "Not part of the MI355X build (see module doc)."})


Beta = _placeholder('Beta')
Exponential = _placeholder('Exponential')
Gamma = _placeholder('Gamma')
Laplace = _placeholder('Laplace')
Poisson = _placeholder('Poisson')
StudentT = _placeholder('StudentT')
FlowDistribution = _placeholder('FlowDistribution')
