"""Encoders producing the sparse operands of the hot path (names mirror reference ``torchsparsegradutils/encoders``)."""

from .pairwise_encoder import PairwiseEncoder, calc_pairwise_coo_indices_nd

__all__ = ["PairwiseEncoder", "calc_pairwise_coo_indices_nd"]
