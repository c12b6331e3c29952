"""Loader for oracle/tt_cpu.c (the plain-C CPU twins of a subset of the C ABI).  TEST INFRASTRUCTURE ONLY.

``load()`` compiles the file with gcc when the shared object is missing or older than the source and returns a ctypes handle
whose functions carry the SAME prototypes as their ``tt_*`` counterparts (taken from ``timetuning_amd._lib.SIGNATURES``), so a
test can drive the HIP library and the twin through one call site, with host pointers on this side."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "tt_cpu.c")
OUT = os.path.join(HERE, "_build", "libtt_cpu.so")
TWINS = ["sinkhorn", "ce_loss_fwd_bwd", "img_resample_h", "img_resample_v", "img_color", "img_box_blur", "confusion_counts",
         "upsample_argmax", "kmeans_assign", "col_moments",
         # round 2: the hot path's row ops and (naive) matrix products
         "linear_fwd", "linear_bwd_data", "linear_bwd_weight", "layernorm_fwd", "l2norm_fwd", "normalize_rows_inplace", "attention_fwd",
         "adamw_step", "ema_update", "queue_push", "scale_rows_inplace", "sinkhorn_from_q", "split_planes", "count_mismatch",
         "colsum", "add_inplace", "layernorm_bwd", "l2norm_bwd", "attention_bwd", "attention_bwd_bf16", "attention_bwd_pairs", "attention_bwd_pairs_workspace_bytes", "patch_embed_fwd", "affine_cols_inplace",
         "transpose_planes", "transpose_planes_colsum", "transpose_planes_colsum_workspace_bytes", "layernorm_fwd_planes", "linear_fwd_planes", "attention_fwd_bf16", "patch_embed_fwd_planes",
         "patch_embed_planes_workspace_bytes",
         # the coarse entry points: the same sequences over the twins
         "vit_forward", "vit_forward_workspace_bytes", "mlp_head_forward", "mlp_head_forward_workspace_bytes", "scores_sinkhorn",
         "scores_sinkhorn_workspace_bytes", "adamw_ema_step",
         # third batch: what had a torch restatement only
         "scale_tensors", "gemm_f32", "pos_embed_interpolate", "upsample_bilinear_tokens", "upsample_argmax_f32", "kmeans_accumulate",
         "kmeans_accumulate_workspace_bytes", "linear_bwd_data_planes", "linear_bwd_weight_planes", "linear_bwd_weight_planes_workspace_bytes",
         "foreground_mask", "foreground_mask_from_probs", "label_propagate", "label_propagate_maps", "label_propagate_workspace_bytes",
         "label_propagate_sims", "label_propagate_from_sims",
         # round 4: the fp16-pair operands of the "f16x3" mode
         "split_pairs", "join_pairs", "layernorm_fwd_pairs", "linear_fwd_pairs", "linear_fwd_pairs_route", "attention_fwd_pairs", "split_pairs_dual",
         "split_pairs_dual_workspace_bytes", "transpose_pairs", "linear_bwd_data_pairs", "linear_bwd_weight_pairs",
         "linear_bwd_weight_pairs_workspace_bytes", "linear_bwd_weight_pairs_tn", "linear_bwd_weight_pairs_tn_ok",
         "linear_bwd_weight_pairs_tn_workspace_bytes", "split_pairs_dual_multi",
         "split_pairs_dual_parts", "linear_bwd_weight_pairs_tn_bias", "amax_slot_bytes",
         "patch_embed_fwd_pairs", "patch_embed_pairs_workspace_bytes",
         "sinkhorn_local_workspace_bytes", "sinkhorn_local_begin", "sinkhorn_local_step", "sinkhorn_local_end"]
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.isfile(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wno-unused-parameter", SRC, "-o", OUT, "-lm"], check=True)
    return OUT


def load():
    global _lib
    if _lib is None:
        from timetuning_amd._lib import SIGNATURES

        lib = C.CDLL(build())
        for name in TWINS:
            fn = getattr(lib, "tt_cpu_" + name)
            fn.restype, fn.argtypes = SIGNATURES["tt_" + name]
        _lib = lib
    return _lib
