"""FiBiNET bilinear interaction layer on MI355X — drop-in for the reference's
`bilinear_interaction_layer(input, embedding_dim, type, name)`
(/root/reference algorithm/FiBiNET/bilinear_interaction_layer.py:5-42).

Same signature, variables (`{name}_w_all` (K,K) | `{name}_w_each` (F-1,K,K) |
`{name}_w_interaction` (F(F-1)/2,K,K)), the same `ValueError` for an unknown type, and the same
pair set `itertools.combinations(range(F-1), 2)` — i.e. (F-1)(F-2)/2 pairs, the last field never
participates, and for "interaction" only the first (F-1)(F-2)/2 weight slices are used (SURVEY.md
quirk B-3).  One HIP kernel (`recalgo_bilinear_fwd/bwd`); `bilinear_interaction_pair` runs the
original and the SENET branch of fibinet.py:177-186 in one launch and writes their concat.
"""
from __future__ import annotations

import torch

from ... import ops
from ...variables import current_store

_TYPES = ("all", "each", "interaction")


def _weight(store, F: int, embedding_dim: int, type: str, name: str):
    K = int(embedding_dim)
    if type == "all":
        return store.get_variable(f"{name}_w_all", (K, K))
    if type == "each":
        return store.get_variable(f"{name}_w_each", (F - 1, K, K))
    if type == "interaction":
        return store.get_variable(f"{name}_w_interaction", (F * (F - 1) // 2, K, K))
    raise ValueError(f"Bilinear Interaction type must be in ['all','each','interaction'], got '{type}'")


def bilinear_interaction_layer(input: torch.Tensor, embedding_dim: int, type: str, name: str) -> torch.Tensor:
    """
    Args:
        input: (batch, F, K)
        embedding_dim: K
        type: "all", "each" or "interaction"
        name: distinguishes the weights of different layers
    Returns:
        (batch, (F-1)(F-2)/2, K)
    """
    store = current_store()
    w = _weight(store, int(input.shape[1]), embedding_dim, type, name)
    return ops.bilinear_interaction(store, type, input.contiguous(), w)


def bilinear_interaction_pair(input0: torch.Tensor, name0: str, input1: torch.Tensor, name1: str,
                              embedding_dim: int, type: str) -> torch.Tensor:
    """concat([bilinear(input0, name0), bilinear(input1, name1)], axis=-1) in one launch:
    (batch, (F-1)(F-2)/2, 2K)."""
    store = current_store()
    F = int(input0.shape[1])
    w0 = _weight(store, F, embedding_dim, type, name0)
    w1 = _weight(store, F, embedding_dim, type, name1)
    return ops.bilinear_interaction(store, type, input0.contiguous(), w0, input1.contiguous(), w1)
