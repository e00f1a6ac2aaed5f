"""Module path of the reference (zhusuan/distributions/laplace.py): the class lives in torch_families.py."""
from .torch_families import Laplace

__all__ = ['Laplace']
