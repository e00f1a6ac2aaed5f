"""Module path of the reference (zhusuan/distributions/poisson.py): the class lives in torch_families.py."""
from .torch_families import Poisson

__all__ = ['Poisson']
