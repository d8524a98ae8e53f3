"""Distributions built on the sparse hot path (names mirror reference ``torchsparsegradutils/distributions``)."""

from .sparse_multivariate_normal import SparseMultivariateNormal

__all__ = ["SparseMultivariateNormal"]
