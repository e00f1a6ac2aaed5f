"""Module path of the reference (zhusuan/distributions/gamma.py): the class lives in torch_families.py."""
from .torch_families import Gamma

__all__ = ['Gamma']
