"""Distributions of the variational-inference hot path (Normal, Bernoulli) and, as the first widening step
(SURVEY.md 8f rank 4), the reference's two other hand-written samplers (Logistic, Uniform).

The reference also ships Beta, Gamma, Laplace, StudentT, Poisson, Exponential (thin wrappers of
torch.distributions) and FlowDistribution (zhusuan/distributions/__init__.py:3-13); they are off the hot path
named by BASELINE.json and are not part of this build."""
from .base import Distribution
from .normal import Normal
from .bernoulli import Bernoulli
from .logistic import Logistic
from .uniform import Uniform
from ._outside import Beta, Exponential, Gamma, Laplace, Poisson, StudentT, FlowDistribution

__all__ = ['Distribution', 'Normal', 'Bernoulli', 'Logistic', 'Uniform',
           # placeholders that raise NotImplementedError when constructed:
           'Beta', 'Exponential', 'Gamma', 'Laplace', 'Poisson', 'StudentT', 'FlowDistribution']
