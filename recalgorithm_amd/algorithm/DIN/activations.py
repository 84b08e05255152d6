"""DIN activations on MI355X — drop-in for the reference's `prelu(x, name)` / `dice(x, name)`
(/root/reference algorithm/DIN/activations.py:4-37).  Same variable names
(`prelu_alpha_{name}`, `dice_alpha_{name}`, per-channel, initialised to 1.0).  Dice keeps the
reference's behaviour of a batch-norm that never trains: p = sigmoid(x / sqrt(1 + 1e-3))
(quirk B-5).  Elementwise HIP kernels `recalgo_activation_fwd/bwd`."""
from __future__ import annotations

import torch

from ... import ops
from ...variables import current_store, ones


def prelu(x: torch.Tensor, name="") -> torch.Tensor:
    store = current_store()
    alpha = store.get_variable(f"prelu_alpha_{name}", (int(x.shape[-1]),), ones)
    return ops.activation(store, x, alpha, "prelu")


def dice(x: torch.Tensor, name="") -> torch.Tensor:
    store = current_store()
    alpha = store.get_variable(f"dice_alpha_{name}", (int(x.shape[-1]),), ones)
    return ops.activation(store, x, alpha, "dice")
