"""Distribution base class.  Interface of zhusuan/distributions/base.py:70-192 of the reference:
``sample(n_samples)``, ``log_prob(given)``, ``prob``, ``batch_shape``, ``dtype``, ``device``,
``is_reparameterized``; subclasses provide ``_sample``, ``_log_prob_sum``, ``_batch_shape``.

Difference in mechanism, not in result: the sum over the last ``group_ndims`` axes
(base.py:175-176) is done inside the log-prob kernel (``_log_prob_sum(given, n_fold)``) instead of
by a separate ``torch.sum`` pass over an element-wise tensor.
"""
import torch

__all__ = ['Distribution']


class Distribution(object):
    # True for families whose NON-reparameterised draw is a function with an identically zero derivative w.r.t. the
    # parameters (Normal: torch.normal(mean, std), normal.py:102; Bernoulli: torch.bernoulli, bernoulli.py:80): the
    # objectives may then hand the draw to the generator detached.  Uniform is NOT one of them: its draw is rescaled
    # outside the no-grad region (uniform.py:63-70) and carries d/d low = 1 - u, d/d high = u.
    _nonreparam_draw_has_zero_grad = False

    def __init__(self,
                 dtype,
                 is_continuous,
                 is_reparameterized,
                 use_path_derivative=False,
                 group_ndims=0,
                 device=None,
                 **kwargs):
        self._dtype = dtype
        self._is_continuous = is_continuous
        self._is_reparameterized = is_reparameterized
        self._use_path_derivative = use_path_derivative
        self._device = device
        self._group_ndims = 0
        if isinstance(group_ndims, int):
            if group_ndims < 0:
                raise ValueError("group_ndims must be non-negative.")
            self._group_ndims = group_ndims
        self.sample_cache = None

    @property
    def dtype(self):
        """The sample type of the distribution."""
        return self._dtype

    @property
    def device(self):
        return self._device

    @property
    def is_continuous(self):
        return self._is_continuous

    @property
    def is_reparameterized(self):
        return self._is_reparameterized

    @property
    def group_ndims(self):
        return self._group_ndims

    @property
    def batch_shape(self):
        return self._batch_shape()

    def _batch_shape(self):
        raise NotImplementedError()

    def sample(self, n_samples=None, **kwargs):
        """base.py:132-150: ``None`` (and 1) give one sample of shape ``batch_shape`` with no leading
        axis; an int K > 1 gives ``[K] + batch_shape``."""
        if n_samples is None:
            return self._sample(n_samples=1, **kwargs)
        elif isinstance(n_samples, int):
            return self._sample(n_samples, **kwargs)
        raise TypeError("n_samples must be None or an int")

    def _sample(self, n_samples, **kwargs):
        raise NotImplementedError()

    def log_prob(self, given):
        """base.py:161-178: log density / mass at `given`, summed over the last `group_ndims` axes."""
        if given is not None:
            given = torch.as_tensor(given, dtype=self.dtype)
        return self._log_prob_sum(given, self._group_ndims)

    def _log_prob_sum(self, given, n_fold):
        """log-prob of `given` (None = the cached sample) summed over the last `n_fold` axes."""
        raise NotImplementedError()

    def _log_prob(self, sample=None, **kwargs):
        """Element-wise log-prob of `sample` (None = the cached sample), no group sum -- the reference's per-family
        hook (normal.py:109, bernoulli.py:84, logistic.py:69, uniform.py:72)."""
        return self._log_prob_sum(sample, 0)

    def prob(self, given):
        return self._prob(given)

    def _prob(self, given):
        return torch.exp(self._log_prob(given))
