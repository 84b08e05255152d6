"""DIN attention on MI355X — drop-in for the reference's
`din_attention(query, keys, keys_length, is_softmax=False)`
(/root/reference algorithm/DIN/din_attention.py:4-43).

Same signature, same variables (`f1_att/{kernel,bias}` (4H,64), `f2_att` (64,32), `f3_att`
(32,1), AUTO_REUSE: created once in the current scope), both branches (softmax with the
`-2**32+1` pad score applied before the 1/sqrt(H) scale; default raw masked scores).  The whole
function is one hand-written HIP kernel (`recalgo_din_attention_fwd/bwd`).
"""
from __future__ import annotations

import torch

from ... import ops
from ...variables import current_store, glorot_uniform, zeros


def din_attention(query: torch.Tensor, keys: torch.Tensor, keys_length: torch.Tensor,
                  is_softmax: bool = False, query_join=None) -> torch.Tensor:
    """
    Args:
        query: target item, (B, H)
        keys: behaviour history, (B, T, H), zero padded
        keys_length: history lengths, (B,)
        is_softmax: softmax-normalise the attention scores
        query_join: (not a reference argument) nn.GradJoin shared with the query's other consumer
    Returns:
        (B, H) weighted sum pooling of the history
    """
    store = current_store()
    H = int(query.shape[-1])
    vs = []
    for name, shape in (("f1_att", (4 * H, 64)), ("f2_att", (64, 32)), ("f3_att", (32, 1))):
        with store.variable_scope(name):
            vs.append(store.get_variable("kernel", shape, glorot_uniform))
            vs.append(store.get_variable("bias", (shape[1],), zeros))
    return ops.din_attention(store, query.contiguous(), keys.contiguous(), keys_length, vs, is_softmax, query_join)
