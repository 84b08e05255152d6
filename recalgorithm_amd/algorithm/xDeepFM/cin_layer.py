"""xDeepFM Compressed Interaction Network layer on MI355X — drop-in for the reference's
`cin_layer(x0, xk, hk_1, index)` (/root/reference algorithm/xDeepFM/cin_layer.py:4-30).

Same signature and variable (`cin_layer_{index}_filter`, shape (1, hk*m, hk_1), glorot-uniform,
no bias, no activation).  The reference materialises the (batch, D, hk*m) outer product and runs
a width-1 conv1d over it; here the layer is one implicit-GEMM HIP kernel on the fp32 matrix
cores (`recalgo_cin_layer_fwd/bwd`) that forms the outer product in registers.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from ... import ops
from ...variables import current_store


def _filter(store, x0, xk, hk_1, index):
    m = int(x0.shape[1])
    hk = int(xk.shape[1])
    return store.get_variable(f"cin_layer_{index}_filter", (1, hk * m, int(hk_1)))


def cin_layer(x0: torch.Tensor, xk: torch.Tensor, hk_1, index: int) -> torch.Tensor:
    """
    Args:
        x0: original input, (batch, m, D)
        xk: previous CIN layer output, (batch, hk, D)
        hk_1: number of feature maps of this layer (int or str — quirk B-2)
        index: layer number (names the filter variable)
    Returns:
        (batch, hk_1, D)
    """
    store = current_store()
    filt = _filter(store, x0, xk, hk_1, index)
    out, _pooled = ops.cin_layer(store, x0.contiguous(), xk.contiguous(), filt)
    return out


def cin_network(x0: torch.Tensor, feature_maps) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """The CIN stack of xdeepfm.py:166-174: every layer's maps are sum-pooled over D and
    concatenated -> p_plus (batch, sum h_i).  The pooling is fused into the layer kernel."""
    store = current_store()
    # every layer within one launch's width: the whole stack is one autograd node (ops.cin_stack)
    filts, hk = [], int(x0.shape[1])
    for i, h in enumerate(feature_maps):
        filts.append(store.get_variable(f"cin_layer_{i + 1}_filter", (1, hk * int(x0.shape[1]), int(h))))
        hk = int(h)
    fused = ops.cin_stack(store, x0, filts)
    if fused is not None:
        return fused[1], fused[0]
    xk, xs, pools = x0, [], []
    for i, h in enumerate(feature_maps):
        filt = _filter(store, x0, xk, h, i + 1)
        xk, pooled = ops.cin_layer(store, x0, xk, filt)
        xs.append(xk)
        pools.append(pooled)
    return xs, (pools[0] if len(pools) == 1 else torch.cat(pools, dim=-1))
