"""Distributions of the variational-inference hot path (Normal, Bernoulli) and, as the first widening step
(SURVEY.md 8f rank 4), the reference's two other hand-written samplers (Logistic, Uniform).

The reference also ships Beta, Gamma, Laplace, StudentT, Poisson, Exponential (thin wrappers of
torch.distributions, zhusuan/distributions/__init__.py:3-13): off the hot path named by BASELINE.json, kept as equally thin
pass-throughs (``torch_families.py``: plain torch ops, no kernels) so that existing model code keeps working.  FlowDistribution
(needs zhusuan.invertible) is not part of this build."""
from .base import Distribution
from .normal import Normal
from .bernoulli import Bernoulli
from .logistic import Logistic
from .uniform import Uniform
from .torch_families import Beta, Exponential, Gamma, Laplace, Poisson, StudentT, FlowDistribution

__all__ = ['Distribution', 'Normal', 'Bernoulli', 'Logistic', 'Uniform',
           # torch.distributions pass-throughs (no kernels) and one placeholder (FlowDistribution raises):
           'Beta', 'Exponential', 'Gamma', 'Laplace', 'Poisson', 'StudentT', 'FlowDistribution']
