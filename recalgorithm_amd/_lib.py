"""ctypes binding of librecalgo_hip.so (the C-ABI declared in include/recalgo.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol
declared in the header is absent, loading raises immediately.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RECALGO_HIP_LIB: a developer switch — A/B runs of differently tuned builds of the same library on one GPU box)
LIB_PATH = os.environ.get("RECALGO_HIP_LIB") or os.path.join(_HERE, "librecalgo_hip.so")

P = c_void_p  # device pointer / stream
ABI_VERSION = 3  # == RECALGO_ABI_VERSION of include/recalgo.h (bumped on any signature change)

# name -> (restype, argtypes); must list every function of include/recalgo.h
SIGNATURES = {
    "recalgo_abi_version": (c_int, []),
    "recalgo_target_arch": (c_char_p, []),
    "recalgo_copy_bytes": (c_int, [P, P, c_int64, P]),
    "recalgo_embedding_gather_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P, c_int, c_int, P]),
    "recalgo_embedding_gather_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "recalgo_scatter_rows_sorted": (c_int, [P, P, P, c_int64, c_int, P, P]),
    "recalgo_embedding_bag_mean_fwd": (c_int, [P, P, P, c_int, c_int, P, c_int, c_int, P]),
    "recalgo_embedding_bag_mean_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "recalgo_sequence_gather_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P]),
    "recalgo_sequence_gather_bwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P]),
    "recalgo_deepfm_sparse_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P]),
    "recalgo_deepfm_sparse_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P]),
    "recalgo_cross_fwd": (c_int, [P, c_int, P, P, c_int, c_int, c_int, P, c_int, P]),
    "recalgo_gather_cross_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, c_int, P, c_int, P, c_int, P]),
    "recalgo_cross_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "recalgo_cross_bwd": (c_int, [P, c_int, P, P, P, c_int, P, c_int, c_int, c_int, P, P, P, P, c_int, P]),
    "recalgo_cross_bwd_partial_rows": (c_int, [c_int]),
    "recalgo_cin_layer_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P]),
    "recalgo_cin_layer_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "recalgo_cin_layer_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      P, c_int, P, c_int, P, P, P]),
    "recalgo_din_attention_fwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "recalgo_din_attention_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "recalgo_din_attention_bwd_partial_rows": (c_int, [c_int]),
    "recalgo_din_attention_bwd_partial_floats": (c_int, [c_int]),
    "recalgo_din_attention_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int,
                                          P, P, P, P, P, P, P, P, P, P]),
    "recalgo_din_attention_bwd_joined": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int,
                                                 P, P, P, P, P, P, P, P, P, P]),
    "recalgo_senet_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "recalgo_senet_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "recalgo_senet_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P]),
    "recalgo_bilinear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_int, c_int, P]),
    "recalgo_bilinear_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "recalgo_bilinear_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "recalgo_pnn_feature_count": (c_int, [c_int, c_int, c_int]),
    "recalgo_pnn_features_fwd": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P]),
    "recalgo_pnn_features_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    "recalgo_pnn_weights_fwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "recalgo_pnn_weights_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "recalgo_mlp_width_supported": (c_int, [c_int]),
    "recalgo_relu_bwd_bias_workspace_bytes": (c_int64, [c_int, c_int]),
    "recalgo_relu_bwd_bias": (c_int, [P, P, c_int, c_int, P, P, P, P]),
    "recalgo_batchnorm_workspace_bytes": (c_int64, [c_int, c_int]),
    "recalgo_batchnorm_train_fwd": (c_int, [P, P, P, c_int, c_int, c_float, c_float, P, P, P, P, P, P, P]),
    "recalgo_batchnorm_train_bwd": (c_int, [P, P, P, P, P, c_int, c_int, P, P, P, P, c_int, P]),
    "recalgo_batchnorm_bwd_act_workspace_bytes": (c_int64, [c_int, c_int]),
    "recalgo_batchnorm_train_bwd_act": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int, P]),
    "recalgo_batchnorm_partial_rows": (c_int, [c_int]),
    "recalgo_batchnorm_moments": (c_int, [P, c_int, c_int, P, P]),
    "recalgo_batchnorm_apply": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P, P, P, P, P]),
    "recalgo_batchnorm_bwd_sums": (c_int, [P, P, P, P, c_int, c_int, P, P]),
    "recalgo_batchnorm_bwd_apply": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, c_int, P]),
    "recalgo_dropout_fwd": (c_int, [P, c_int64, c_double, P, c_int, c_int, P, P, P]),
    "recalgo_dropout_bwd": (c_int, [P, c_int64, c_double, P, c_int, c_int, P, P, P]),
    "recalgo_dropout_keep_mask": (c_int, [c_int64, c_double, c_int, c_int, P, P, P]),
    "recalgo_dense_fwd_drop": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, P, c_int, P, P, P]),
    "recalgo_batchnorm_apply_drop": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P, P, P, P, P, P]),
    "recalgo_batchnorm_train_bwd_drop": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int, c_float, P, P]),
    "recalgo_sigmoid_ce_fwd_bwd": (c_int, [P, P, c_int, c_float, P, P, P, P]),
    "recalgo_adam_tf1_dense": (c_int, [P, P, P, P, c_int64, c_float, P, c_float, c_float, c_float, c_int, P]),
    "recalgo_adam_tf1_rows": (c_int, [P, P, P, P, P, c_int64, c_int, c_float, P, c_float, c_float, c_float, c_int, P]),
    "recalgo_mark_live_rows": (c_int, [P, P, c_int64, c_int, P, P, P, P]),
    "recalgo_dense1_fwd": (c_int, [P, P, c_int, c_int, P, P, P, P]),
    "recalgo_dense1_bwd_workspace_bytes": (c_int64, [c_int, c_int]),
    "recalgo_dense1_bwd": (c_int, [P, P, c_int, c_int, P, P, P, P, P, P, P]),
    "recalgo_order_live_list_workspace_bytes": (c_int64, [c_int64]),
    "recalgo_order_live_list": (c_int, [P, c_int64, P, P, P, P]),
    "recalgo_dedup_rows_workspace_bytes": (c_int64, [c_int64]),
    "recalgo_dedup_rows": (c_int, [P, c_int64, P, P, P, P]),
    "recalgo_exchange_plan": (c_int, [P, c_int64, c_int, c_int64, P, P, P, P, P, P]),
    "recalgo_adam_tf1_list": (c_int, [P, P, P, P, P, P, c_int64, c_int, c_float, P, c_float, c_float, c_float, c_int, P]),
    "recalgo_adam_tf1_step": (c_int, [P, P, P, P, c_int64, P, c_int, P, P, c_int, c_float, c_float, c_float, c_float, c_int, P]),
    "recalgo_adam_tf1_step_plans": (c_int, [P, P, P, P, c_int64, P, c_int, P, P, c_int, c_float, c_float, c_float, c_float, c_int,
                                            P, c_int, P]),
    "recalgo_scatter_plan_scan": (c_int, [P, c_int64, c_int, P]),
    "recalgo_adam_tf1_advance": (c_int, [P, c_float, c_float, c_float, P, P]),
    "recalgo_cross_layer_fwd": (c_int, [P, P, c_int, P, P, c_int, c_int, P, c_int, P]),
    "recalgo_cross_layer_bwd": (c_int, [P, P, c_int, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P]),
    "recalgo_dense_fwd": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, P, c_int, P]),
    "recalgo_dense_fwd_bn": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, P, c_int, P, P]),
    "recalgo_dense_fwd_act_bn": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P]),
    "recalgo_dense_bwd_input": (c_int, [P, c_int, P, P, c_int, c_int, c_int, P, c_int, c_float, P, c_int, c_int, P]),
    "recalgo_dense_bwd_weights_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "recalgo_dense_bwd_weights": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, P, P, P, c_int, P]),
    "recalgo_dense_bwd": (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, P, c_int, c_float, P, c_int, P, P, P, c_int, P]),
    "recalgo_dense_bwd_bn": (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, P, c_int, c_float, P, c_int, P, P, P, c_int,
                                     P, P, P, P, P, c_int, P]),
    "recalgo_dense_bwd_rider_supported": (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int,
                                                  c_int, c_int]),
    "recalgo_dense_bwd_cross_rider_supported": (c_int, [c_int, c_int]),
    "recalgo_dense_bwd_rider": (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, P, c_int, c_float, P, c_int, P, P, P, c_int,
                                        P, P, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, P, P, P,
                                        P, c_int, P, P, P, c_int, c_int, c_int, P, P, P]),
    "recalgo_dense_bwd_weights_reduce": (c_int, [P, c_int, P, c_int, P, P]),
    "recalgo_logit_loss_partial_rows": (c_int64, [c_int]),
    "recalgo_logit_loss_fwd_bwd": (c_int, [P, P, P, c_int, P, P, P, P, P, c_int, c_float, P, P, P, P, P, P, P]),
    "recalgo_tail_partial_rows": (c_int, [c_int]),
    "recalgo_tail_dense_head_supported": (c_int, [c_int, c_int, c_int]),
    "recalgo_tail_dense_head_fwd_bwd": (c_int, [P, c_int, P, P, c_int, P, c_int, c_int, P, P, P, P, P, c_int, c_float,
                                                P, P, P, P, P, P, P, P]),
    "recalgo_bi_interaction_fwd": (c_int, [P, c_int, c_int, c_int, P, P]),
    "recalgo_bi_interaction_bwd": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "recalgo_attention_pool_fwd": (c_int, [P, P, c_int, c_int, c_int, P, P, P]),
    "recalgo_attention_pool_bwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P]),
    "recalgo_ffm_pairs_fwd": (c_int, [P, c_int, c_int, c_int, P, P]),
    "recalgo_ffm_pairs_bwd": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "recalgo_activation_fwd": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "recalgo_activation_bwd_workspace_bytes": (c_int64, [c_int, c_int]),
    "recalgo_activation_bwd_partial_rows": (c_int, [c_int, c_int]),
    "recalgo_activation_bwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P]),
    "recalgo_embedding_gather_fwd_deferred": (c_int, [P, P, P, c_int, c_int, c_int, P, c_int, c_int, P, P, c_int, P]),
    "recalgo_embedding_bag_mean_fwd_deferred": (c_int, [P, P, P, c_int, c_int, P, c_int, c_int, P, c_int64, P, c_int, P]),
    "recalgo_sequence_gather_fwd_deferred": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, c_int64, P, c_int, P]),
    "recalgo_deepfm_sparse_fwd_deferred": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int, P]),
    "recalgo_concat_sumsq_workspace_bytes": (c_int64, [c_int]),
    "recalgo_concat_sumsq": (c_int, [P, P, c_int, c_int, P, c_float, P, P, P]),
    "recalgo_scatter_plan_buckets_log2": (c_int, [c_int64]),
    "recalgo_scatter_plan_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "recalgo_scatter_source_slots": (c_int64, [c_int, c_int, c_int]),
    "recalgo_scatter_plan_header_bytes": (c_int64, [c_int]),
    "recalgo_scatter_prepare": (c_int, [P, c_int, P, c_int64, c_int, c_int64, c_int, P, P, c_int64, c_int64, c_int, P, c_int, P]),
    "recalgo_lookup_multi_fwd": (c_int, [P, c_int, P]),
    "recalgo_scatter_prepare_multi": (c_int, [P, c_int, P, c_int, P, c_int64, c_int, c_int, P, c_int64, c_int, P, c_int, P]),
    "recalgo_scatter_apply": (c_int, [P, c_int, P, c_int, P, c_int64, c_int, c_int, P, P, P, P, P, c_int64, P, P, c_int,
                                      c_float, c_float, c_float, c_float, P]),
    "recalgo_adam_deferred_sweep": (c_int, [P, c_int, c_int64, c_int64, P, c_int, P]),
}

_lib = None


class RecalgoError(RuntimeError):
    pass


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load the HIP library (once).  Import torch first so that the process-wide HIP
    runtime (libamdhip64.so.7) is the one torch already mapped."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RecalgoError(
            f"{path} not found: build it with `python -m recalgorithm_amd.build` "
            "(there is no CPU fallback for the hot path)")
    import torch  # noqa: F401  (maps torch's libamdhip64 before ours resolves its NEEDED)
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RecalgoError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.recalgo_abi_version() != ABI_VERSION:
        raise RecalgoError(f"{path}: ABI version {lib.recalgo_abi_version()}, this binding expects {ABI_VERSION} "
                           "(a stale build: python -m recalgorithm_amd.build)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RecalgoError(f"{what} failed with hipError_t={rc}")
