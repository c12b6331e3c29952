"""ctypes binding of libtimetuning_hip.so (the C ABI in include/timetuning_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing a HIP op raises.  ``build()`` compiles the library in-tree with hipcc
(gfx950); the resulting ``timetuning_amd/libtimetuning_hip.so`` travels to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TT_LIB_PATH: measurement tools only - a variant build of the library, tools/build_variant.sh)
LIB_PATH = os.environ.get("TT_LIB_PATH") or os.path.join(_HERE, "libtimetuning_hip.so")
CSRC = os.path.join(_HERE, "csrc")

c_f32p = C.c_void_p
c_vp = C.c_void_p
c_i = C.c_int
c_ll = C.c_longlong
c_f = C.c_float
c_d = C.c_double
c_sz = C.c_size_t


class AdamwTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_longlong),
                ("lr", C.c_float), ("weight_decay", C.c_float)]


class VitBlockParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b",
                                          "fc2_w", "fc2_b", "qkv_wp", "proj_wp", "fc1_wp", "fc2_wp")]


class VitParams(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("cls", C.c_void_p), ("pos", C.c_void_p),
                ("blocks", C.POINTER(VitBlockParams)), ("n_blocks", C.c_int), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
                ("dim", C.c_int), ("heads", C.c_int), ("hidden", C.c_int), ("patch", C.c_int), ("planes", C.c_int), ("patch_wp", C.c_void_p), ("range_flag", C.c_void_p),
                ("precision", C.c_int)]


class LinearParams(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("out_features", C.c_int), ("in_features", C.c_int)]


# name -> (restype, argtypes); must list every symbol include/timetuning_hip.h declares
SIGNATURES = {
    "tt_last_error": (C.c_char_p, []),
    "tt_abi_version": (c_i, []),
    "tt_set_tuning_knob": (c_i, [C.c_char_p, c_i]),
    "tt_device_info": (c_i, [C.c_char_p, c_i]),
    "tt_linear_ksplit_workspace_bytes": (c_sz, []),
    "tt_linear_ksplit_workspace_init": (c_i, [c_vp, c_sz, c_vp]),
    "tt_linear_fwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp]),   # (..., act, precision, stream)
    "tt_linear_bwd_data": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp]),
    "tt_linear_bwd_weight": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_linear_bwd_weight_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "tt_linear_bwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_colsum_workspace_bytes": (c_sz, [c_i, c_i]),
    "tt_colsum": (c_i, [c_vp, c_vp, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_gemm_f32": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_ll, c_ll, c_ll, c_vp]),
    "tt_gemm_tile_choice": (c_i, [c_i, c_i, c_i]),
    "tt_linear_fwd_route": (c_i, [c_i, c_i, c_i]),
    "tt_linear_fwd_planes_route": (c_i, [c_i] * 10),
    "tt_patch_embed_fwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "tt_patch_embed_planes_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "tt_patch_embed_fwd_planes": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_layernorm_fwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_f, c_i, c_vp]),
    "tt_layernorm_bwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp, c_vp]),
    "tt_layernorm_bwd_workspace_bytes": (c_sz, [c_i, c_i]),
    "tt_attention_fwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp]),
    "tt_split_planes": (c_i, [c_vp, c_vp, c_ll, c_i, c_ll, c_vp]),
    "tt_layernorm_fwd_planes": (c_i, [c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_vp, c_vp, c_i, c_i, c_f, c_i, c_vp]),
    "tt_linear_fwd_planes": (c_i, [c_vp, c_ll, c_vp, c_ll, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_attention_fwd_bf16": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp]),
    "tt_split_pairs": (c_i, [c_vp, c_vp, c_ll, c_vp, c_vp]),
    "tt_join_pairs": (c_i, [c_vp, c_vp, c_ll, c_vp]),
    "tt_layernorm_fwd_pairs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_f, c_i, c_vp, c_vp]),
    "tt_linear_fwd_pairs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp, c_vp]),
    "tt_linear_fwd_pairs_route": (c_i, [c_i] * 9),
    "tt_attention_fwd_pairs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp]),
    "tt_split_pairs_dual_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "tt_split_pairs_dual": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp, c_vp]),
    "tt_amax_slot_bytes": (c_sz, []),
    "tt_split_pairs_dual_parts": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp, c_vp]),
    "tt_transpose_pairs": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_vp]),
    "tt_linear_bwd_data_pairs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp, c_vp]),
    "tt_linear_bwd_weight_pairs_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "tt_linear_bwd_weight_pairs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_split_pairs_dual_multi": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_vp]),
    "tt_patch_embed_pairs_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "tt_patch_embed_fwd_pairs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp, c_vp]),
    "tt_linear_bwd_weight_pairs_tn_ok": (c_i, [c_i, c_i, c_i]),
    "tt_linear_bwd_weight_pairs_tn_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "tt_linear_bwd_weight_pairs_tn": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_linear_bwd_weight_pairs_tn_bias": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp, c_i, c_vp, c_vp]),
    "tt_transpose_planes": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_vp]),
    "tt_transpose_planes_colsum_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "tt_transpose_planes_colsum": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_vp, c_vp, c_sz, c_vp]),
    "tt_linear_bwd_data_planes": (c_i, [c_vp, c_ll, c_vp, c_ll, c_i, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_linear_bwd_weight_planes": (c_i, [c_vp, c_ll, c_vp, c_ll, c_i, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_linear_bwd_weight_planes_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "tt_attention_bwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp, c_sz, c_vp, c_vp]),
    "tt_attention_bwd_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "tt_attention_bwd_bf16": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp, c_sz, c_vp]),
    "tt_attention_bwd_pairs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "tt_attention_bwd_pairs_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "tt_l2norm_fwd": (c_i, [c_vp, c_i, c_vp, c_vp, c_i, c_i, c_vp]),
    "tt_l2norm_bwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_vp, c_vp]),
    "tt_normalize_rows_inplace": (c_i, [c_vp, c_i, c_i, c_vp]),
    "tt_sinkhorn": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_i, c_vp, c_sz, c_vp]),
    "tt_sinkhorn_workspace_bytes": (c_sz, [c_i, c_i]),
    "tt_sinkhorn_local_workspace_bytes": (c_sz, [c_i, c_i]),
    "tt_sinkhorn_local_begin": (c_i, [c_vp, c_vp, c_i, c_i, c_f, c_vp, c_sz, c_vp]),
    "tt_sinkhorn_local_step": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_sinkhorn_local_end": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_sinkhorn_from_q": (c_i, [c_vp, c_i, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_label_propagate": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_vp, c_sz, c_vp]),
    "tt_label_propagate_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "tt_ce_loss_fwd_bwd": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_f, c_vp, c_sz, c_vp]),
    "tt_ce_workspace_bytes": (c_sz, [c_i]),
    "tt_queue_push": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp]),
    "tt_adamw_step": (c_i, [C.POINTER(AdamwTensor), c_i, c_i, c_f, c_f, c_f, c_vp]),
    "tt_scale_tensors": (c_i, [C.POINTER(AdamwTensor), c_i, c_vp, c_vp]),
    "tt_ema_update": (c_i, [c_vp, c_vp, c_ll, c_d, c_vp]),
    "tt_vit_forward_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "tt_vit_forward": (c_i, [C.POINTER(VitParams), c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tt_mlp_head_forward_workspace_bytes": (c_sz, [c_i, C.POINTER(LinearParams), c_i]),
    "tt_mlp_head_forward": (c_i, [c_vp, c_i, C.POINTER(LinearParams), c_i, c_vp, c_i, c_vp, c_sz, c_vp]),
    "tt_scores_sinkhorn_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "tt_scores_sinkhorn": (c_i, [c_vp, c_i, c_vp, c_i, c_vp, c_i, c_i, c_vp, c_vp, c_i, c_f, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_adamw_ema_step": (c_i, [C.POINTER(AdamwTensor), c_i, c_i, c_f, c_f, c_f, c_vp, c_i, c_i, c_vp, c_vp, c_ll, c_vp, c_d, c_vp]),
    "tt_add_inplace": (c_i, [c_vp, c_vp, c_ll, c_vp]),
    "tt_count_mismatch": (c_i, [c_vp, c_vp, c_ll, c_vp, c_vp]),
    "tt_foreground_mask": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_vp]),
    "tt_foreground_mask_from_probs": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_vp]),
    "tt_scale_rows_inplace": (c_i, [c_vp, c_vp, c_i, c_i, c_vp]),
    "tt_pos_embed_interpolate": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_f, c_vp]),
    "tt_img_resample_h": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "tt_img_resample_v": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.POINTER(C.c_float), C.POINTER(C.c_float), c_vp]),
    "tt_img_color": (c_i, [c_vp, c_i, c_i, c_i, c_i, c_f, c_i, c_vp, c_vp]),
    "tt_img_box_blur": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, C.c_uint, C.c_uint, c_vp]),
    "tt_affine_cols_inplace": (c_i, [c_vp, c_vp, c_vp, c_ll, c_i, c_vp]),
    "tt_col_moments_workspace_bytes": (c_sz, [c_ll, c_i]),
    "tt_col_moments": (c_i, [c_vp, c_vp, c_vp, c_ll, c_i, c_vp, c_sz, c_vp]),
    "tt_upsample_bilinear_tokens": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp]),
    "tt_upsample_argmax_f32": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp]),
    "tt_kmeans_assign": (c_i, [c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_vp]),
    "tt_kmeans_accumulate_workspace_bytes": (c_sz, [c_ll, c_i, c_i]),
    "tt_kmeans_accumulate": (c_i, [c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_label_propagate_sims": (c_i, [c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "tt_label_propagate_from_sims": (c_i, [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_vp, c_sz, c_vp]),
    "tt_label_propagate_maps": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_vp, c_sz, c_vp]),
    "tt_upsample_argmax": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp]),
    "tt_confusion_counts": (c_i, [c_vp, c_vp, c_ll, c_i, c_vp, c_vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def build(verbose: bool = False, force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into timetuning_amd/libtimetuning_hip.so (hipcc cross-compiles without a GPU).
    ``force`` recompiles every source (``make -B``): what ``__graft_entry__.build()`` uses, so that object files left in the
    working tree cannot stand in for a source that no longer compiles."""
    proc = subprocess.run(["make", "-C", CSRC, "-j8"] + (["-B"] if force else []), capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout[-4000:])
        print(proc.stderr[-4000:])
    if proc.returncode != 0:
        raise HipLibraryError("building libtimetuning_hip.so failed")
    return LIB_PATH


def load():
    """Return the loaded library with argtypes set; raises HipLibraryError (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or `make -C timetuning_amd/csrc`). There is no CPU fallback for the HIP path.")
    # torch first: it brings ITS libamdhip64 into the process.  Loaded before torch, this library pulls /opt/rocm's copy in, the process
    # then holds two HIP runtimes and the first kernel launch fails with "no ROCm-capable device is detected" (build() followed by
    # smoke() in one interpreter, round 4).  The streams and buffers every entry point receives come from torch anyway.
    import torch  # noqa: F401

    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tt_last_error().decode(errors="replace")
        raise HipLibraryError(f"{what} failed (code {rc}): {msg}")
