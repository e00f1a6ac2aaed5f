"""Model-definition layer of the hot path: ``BayesianNet`` (named stochastic nodes on an ``nn.Module``) and
``StochasticTensor`` (a node's distribution, value and reduced log-probability)."""
from .stochastic_tensor import StochasticTensor
from .bn import BayesianNet

__all__ = ['BayesianNet', 'StochasticTensor']
