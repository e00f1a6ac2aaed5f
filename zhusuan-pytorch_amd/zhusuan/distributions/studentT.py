"""Module path of the reference (zhusuan/distributions/studentT.py): the class lives in torch_families.py."""
from .torch_families import StudentT

__all__ = ['StudentT']
