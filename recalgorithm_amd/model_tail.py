"""The prediction / loss / metric / train_op tail every reference model_fn repeats
(e.g. /root/reference algorithm/DeepFM/deepfm.py:214-273): sigmoid, mean sigmoid-CE
(fused HIP kernel, a14), tf.metrics accuracy@0.5 + 200-bucket AUC, Adam(lr, .9, .999, 1e-8)."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import ops
from .estimator import AdamOptimizer, EstimatorSpec, LazyAdamOptimizer, ModeKeys, metrics
from .variables import current_store


def sigmoid(logit: torch.Tensor) -> torch.Tensor:
    return torch.sigmoid(logit)


def finish_model_fn(mode, logit: torch.Tensor, labels, params,
                    predictions: Optional[Callable[[torch.Tensor], Dict[str, torch.Tensor]]] = None,
                    extra_loss: Optional[Callable[[], torch.Tensor]] = None,
                    label_key: str = "read_comment") -> EstimatorSpec:
    from .nn import LazyLogit
    lazy = logit if isinstance(logit, LazyLogit) else None
    extra = None
    tail = lazy.tail() if (lazy is not None and mode == ModeKeys.TRAIN and not current_store().building) else None
    if tail is not None and extra_loss is not None:
        extra = extra_loss()
        if extra is not None and extra.requires_grad:
            tail = None
    if lazy is not None and tail is None:
        ok = mode == ModeKeys.TRAIN and lazy.fusable() and not current_store().building
        if ok and extra_loss is not None:
            # the fused tail delivers the loss VALUE through the step's deferred sums: an extra term can only join it as
            # a detached addend (DIN's regulariser value; its gradient rides on the first fcn layer)
            extra = extra_loss() if extra is None else extra
            ok = extra is None or not extra.requires_grad
        if not ok:
            logit, lazy = lazy.materialize(), None
    if mode == ModeKeys.PREDICT:
        prob = torch.sigmoid(logit)
        preds = predictions(prob) if predictions else {"probabilities": prob}
        return EstimatorSpec(mode, predictions=preds, export_outputs={"prediction": preds})

    y = labels[label_key]
    if tail is not None:
        # one launch: the last hidden layer + the one-unit head over [side, layer] + sigmoid-CE + the backward of all of it
        side, ld, side_first = tail
        head_kernel, head_bias, _ = lazy.heads[0]
        loss, prob, logit = ops.tail_dense_head(current_store(), y, head_kernel, head_bias, ld.kernel, ld.bias, ld.x, side,
                                                side_first, loss_addend=extra)
    elif lazy is not None:
        # one launch: one-unit head(s) + sigmoid-CE + the backward of both (the loss-gradient seed is known)
        heads = [(k, len(ps)) for k, _, ps in lazy.heads]
        bias = next((b for _, b, _ in lazy.heads if b is not None), None)
        parts = [t for _, _, ps in lazy.heads for t in ps]
        loss, prob, logit = ops.logit_loss(current_store(), y, heads, bias, parts, lazy.tensors, loss_addend=extra)
    elif current_store().building:          # variable-registration pass: nothing is launched
        loss, prob = logit.new_zeros(()), torch.zeros_like(logit)
    else:
        loss, prob = ops.sigmoid_cross_entropy(logit, y)
    if extra_loss is not None and lazy is None:
        if extra is None:
            extra = extra_loss()
        if extra is not None:
            loss = loss + extra
    if mode == ModeKeys.EVAL:
        acc = metrics.accuracy(labels=y, predictions=(prob >= 0.5).to(torch.float32))
        auc = metrics.auc(labels=y, predictions=prob)
        return EstimatorSpec(mode, loss=loss, eval_metric_ops={"eval_accuracy": acc, "eval_auc": auc})

    # TRAIN: the reference also builds the two metric ops here, but only to feed tf.summary /
    # LoggingTensorHook (deepfm.py:237-238,256-271); they are not part of the training step
    assert mode == ModeKeys.TRAIN
    opt_cls = LazyAdamOptimizer if params.get("lazy_adam") else AdamOptimizer      # lazy_adam: labelled deviation (§8f-1)
    optimizer = opt_cls(learning_rate=params["learning_rate"], beta1=0.9, beta2=0.999, epsilon=1e-8)
    train_op = optimizer.minimize(loss=loss)
    return EstimatorSpec(mode, loss=loss, train_op=train_op, predictions={"probabilities": prob})
