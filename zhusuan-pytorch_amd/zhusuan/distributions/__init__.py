"""Distributions on the variational-inference hot path (Normal, Bernoulli).

The reference also ships Logistic, Beta, Gamma, Laplace, Uniform, StudentT, Poisson, Exponential
and FlowDistribution (zhusuan/distributions/__init__.py:3-13); they are off the hot path named by
BASELINE.json and are not part of this build."""
from .base import Distribution
from .normal import Normal
from .bernoulli import Bernoulli

__all__ = ['Distribution', 'Normal', 'Bernoulli']
