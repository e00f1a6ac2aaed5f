"""Module path of the reference (zhusuan/distributions/beta.py): the class lives in torch_families.py."""
from .torch_families import Beta

__all__ = ['Beta']
