"""DCN cross layer on MI355X — drop-in for the reference's `cross_layer(x0, xl, index)`
(/root/reference algorithm/DCN/cross_layer.py:4-26).

Same name, argument meaning and variable names (`wl_{index}`, `bl_{index}`, shape (d, 1),
default glorot-uniform initialiser for BOTH — SURVEY.md A-7); the arithmetic
    out = x0 * (xl @ wl) + bl^T + xl
runs in the hand-written HIP kernel `recalgo_cross_layer_fwd/bwd` (include/recalgo.h).
`cross_network` is the fused form used by dcn_model_fn: all L layers in one kernel launch,
with wl_i / bl_i allocated as rows of one [L, d] block (names unchanged).
"""
from __future__ import annotations

import torch

from ... import ops
from ...variables import current_store


def cross_layer(x0: torch.Tensor, xl: torch.Tensor, index: int) -> torch.Tensor:
    """
    Args:
        x0: original cross-network input, (batch, d)
        xl: previous cross layer output, (batch, d)
        index: layer number (names the variables)
    Returns:
        (batch, d)
    """
    store = current_store()
    d = int(x0.shape[-1])
    wl = store.get_variable(f"wl_{index}", (d, 1))
    bl = store.get_variable(f"bl_{index}", (d, 1))
    return ops.cross_layer(store, x0.contiguous(), xl.contiguous(), wl, bl)


def cross_network(x0: torch.Tensor, num_cross_layer: int, grad_join=None) -> torch.Tensor:
    """x_{l+1} = cross_layer(x0, x_l, l) for l = 0..L-1 (dcn.py:157-160), fused.  `grad_join` (nn.GradJoin): shared with
    the other consumer of x0 so that the two input gradients are summed inside the backward kernel."""
    store = current_store()
    d = int(x0.shape[-1])
    L = int(num_cross_layer)
    if L == 0:
        return x0
    w, _ = store.get_variable_block("wl", [f"wl_{i}" for i in range(L)], (d, 1))
    b, _ = store.get_variable_block("bl", [f"bl_{i}" for i in range(L)], (d, 1))
    # outside the fused kernel's envelope — more than 6 layers, d > 1024, or d > 512 with >= 4 fused layers (its backward
    # would spill registers, profiles/r01_kernel_resource_usage.md): layer by layer with the single-layer kernels
    if L > 6 or d > 1024 or (d > 512 and L >= 4):
        ops.flush_lazy_gathers()                  # (a gather left to the fused kernel runs on its own after all)
        if grad_join is not None:
            grad_join.consumer_done = True        # no fused consumer: the other branch returns its gradient normally
        xl = x0
        for i in range(L):
            xl = cross_layer(x0, xl, i)
        return xl
    return ops.cross_stack(store, x0.contiguous(), w, b, grad_join)
