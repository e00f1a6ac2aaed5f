"""Module path of the reference (zhusuan/distributions/exponential.py): the class lives in torch_families.py."""
from .torch_families import Exponential

__all__ = ['Exponential']
