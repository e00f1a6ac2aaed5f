"""Variational objectives of the hot path: the evidence lower bound (sgvb / reinforce estimators) and the
importance-weighted bound (sgvb / vimco).  Import surface of the reference's ``zhusuan.variational`` package."""
from .importance_weighted_objective import ImportanceWeightedObjective
from .elbo import EvidenceLowerBoundObjective, ELBO

__all__ = ['ELBO', 'EvidenceLowerBoundObjective', 'ImportanceWeightedObjective']
