""" Core functionality (BayesianNet, StochasticTensor) """
from .bn import *
from .stochastic_tensor import *
