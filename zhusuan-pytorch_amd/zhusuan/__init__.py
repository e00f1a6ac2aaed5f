"""zhusuan -- MI355X (gfx950) build of ZhuSuan-PyTorch's variational-inference hot path.

Same import surface as the reference package for that path (zhusuan/__init__.py:1-7):
``zhusuan.distributions``, ``zhusuan.framework`` and ``zhusuan.log_mean_exp`` are pulled in here;
``zhusuan.variational`` is imported explicitly by callers
(``from zhusuan.variational.elbo import ELBO``), exactly as with the reference.

The arithmetic of Normal / Bernoulli sampling + log-prob, the K-particle reductions and the
ELBO / IWAE / VIMCO estimators runs in hand-written HIP kernels (zhusuan-pytorch_amd/csrc,
C ABI in include/zs_hip.h).  There is no CPU path: tensors must live on the GPU.
"""
__version__ = '0.0.1+mi355x'

from . import distributions
from . import framework
from .utils import *
from ._rng import inject_epsilon, DeviceRNG, device_rng, reference_rng, pair_draws
from .graph import GraphedStep, GraphedStages
from .framework.stochastic_tensor import skip_discarded_draws
from .layers import particle_linear, particle_mlp, particle_rmse, Linear, Sequential
from . import optim
