"""FiBiNET SENET layer on MI355X — drop-in for the reference's
`senet(input, embedding_dim, reduction_ratio)` (/root/reference algorithm/FiBiNET/senet.py:4-36).

Same signature, variables (`senet_w1` (F, K//r), `senet_w2` (K//r, F), glorot-uniform, no bias)
and the same assertion: the reduction dimension is derived from the embedding dimension K, not
from the number of fields F.  One HIP kernel (`recalgo_senet_fwd/bwd`).
"""
from __future__ import annotations

import torch

from ... import ops
from ...variables import current_store


def senet(input: torch.Tensor, embedding_dim: int, reduction_ratio: int) -> torch.Tensor:
    """
    Args:
        input: (batch, F, K)
        embedding_dim: K
        reduction_ratio: reduction ratio
    Returns:
        (batch, F, K)
    """
    F = int(input.shape[1])
    reduction_dim = int(embedding_dim) // int(reduction_ratio)
    assert reduction_dim < embedding_dim, "reduction_dim must be less than embedding_dim"
    store = current_store()
    w1 = store.get_variable("senet_w1", (F, reduction_dim))
    w2 = store.get_variable("senet_w2", (reduction_dim, F))
    return ops.senet(store, input.contiguous(), w1, w2)
