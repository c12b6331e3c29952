"""Torch-tensor front end of the C ABI (include/timetuning_hip.h).

PyTorch supplies device memory and the HIP stream only; all arithmetic happens in
libtimetuning_hip.so.  Every function validates device / dtype / contiguity and raises if
the library is unavailable - there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib

f32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=f32) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.HipLibraryError(f"{name}: expected a tensor in GPU memory (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ---- ABI 7: the K-split workspace of the persistent GEMMs is the CALLER's (tt_linear_ksplit_workspace_bytes / _init).  This front end keeps
# one per (device, stream) - launches on one stream are ordered, so they can share the partials and the self-resetting counters - allocated
# and initialised at its first use, OUTSIDE any launch path of the library (under a hipGraph capture of a step the buffer already exists:
# ``ksplit_workspace()`` is called once before capturing).
_KSPLIT_WS: dict = {}


def ksplit_workspace(device=None) -> torch.Tensor:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
    ws = _KSPLIT_WS.get(key)
    if ws is None:
        lib = _lib.load()
        with torch.cuda.device(key[0]):
            nb = int(lib.tt_linear_ksplit_workspace_bytes())
            ws = torch.empty(nb, dtype=torch.uint8, device=torch.device("cuda", key[0]))
            _lib.check(lib.tt_linear_ksplit_workspace_init(_p(ws), nb, _stream()), "tt_linear_ksplit_workspace_init")
        _KSPLIT_WS[key] = ws
    return ws


class PairRangeError(_lib.HipLibraryError):
    """An operand of the "f16x3" mode left fp16's range (|x| > 65504) or was not finite: the one way the mode differs from fp32."""


# The pair producers' range flag (include/timetuning_hip.h, "RANGE FLAG"): one device int per device, passed to every entry point that
# splits fp32 values into pairs.  ``check_pair_range`` reads it (a synchronisation: call it where the host synchronises anyway - the
# training driver does when it prints the loss, TimeT.train_update does every RANGE_CHECK_EVERY steps) and raises PairRangeError.
_RANGE_FLAG: dict = {}
RANGE_CHECK = True


def range_flag(device=None) -> Optional[torch.Tensor]:
    if not RANGE_CHECK:
        return None
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index if torch.device(device).index is not None
                                                              else torch.cuda.current_device())
    f = _RANGE_FLAG.get(idx)
    if f is None:
        f = _RANGE_FLAG[idx] = torch.zeros(4, dtype=torch.int32, device=torch.device("cuda", idx))
    return f


def check_pair_range(device=None, reset: bool = True) -> None:
    """Raises PairRangeError if any pair producer since the last check saw a value beyond fp16's range.  Synchronises."""
    f = range_flag(device)
    if f is None:
        return
    bad = int(f[0].item())
    if bad:
        if reset:
            f.zero_()
        raise PairRangeError('an operand of the "f16x3" arithmetic is beyond fp16\'s range (|x| > 65504) or not finite; its products are '
                             "inf / NaN where fp32 arithmetic would not be.  Use --precision f32 (hip_ops.set_gemm_precision('f32')) "
                             "for this model / input.")


# name -> (mode of the C-side switch for tt_linear_fwd, bf16 planes used by the launch sequences for blocks that keep nothing)
# (2 "planes" = fp16 PAIRS: hi / lo halves of an fp32 value, the "f16x3" mode)
_PRECISIONS = {"f32": (0, 0), "bf16x3": (1, 0), "bf16": (2, 1), "bf16x6": (0, 3), "f16x3": (0, 2)}
_precision = "f32"


def set_gemm_precision(mode: str) -> None:
    """Arithmetic of the forward nn.Linear products: a setting of THIS front end (which launch sequences it strings together, and the
    ``precision`` argument it hands to the entry points that take one - ABI 8: the library itself keeps no such state).
    "f32"     (default) exact fp32 MFMA - every headline / parity number.
    "f16x3"   fp32-ACCURATE split mode (round 4): operands pre-split by their producers into fp16 PAIRS (hi = fp16(x), lo = fp16((x - hi)
              2^11): 23 significant bits), three fp16 MFMAs per product term into two fp32 accumulators (gemm_pairs8.hip).  Same blocks
              as "bf16x6" below; per-op error at or under the f32-MFMA kernels' own.
    "bf16x6"  fp32-ACCURATE split mode: operands pre-split by their producers into three bf16 planes (24 significant bits), six
              bf16 MFMAs per product term (gemm_planes.hip).  Applies to the blocks that keep no activations (frozen blocks,
              teacher, the non-target frames of the trainable blocks); everything with a backward stays exact fp32.
    "bf16"    BASELINE C4's MFMA bf16 path: the same blocks run on bf16 activations and weights in HBM (one plane, bf16
              attention); the blocks that keep activations convert fp32 operands while staging (gemm_nt_bf16.hip).
    "bf16x3"  two-plane split (~2^-16 per product) converted while staging; fp32 in HBM."""
    global _precision
    if mode not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
    _lib.load()            # (a missing library fails here, loudly, as it always has)
    _precision = mode


def set_tuning_knob(name: str, value: int) -> None:
    """A dispatcher tuning knob (``TT_<NAME>``; the library reads them once from the environment): A/B tools and tests only."""
    _lib.check(_lib.load().tt_set_tuning_knob(name.encode(), int(value)), "tt_set_tuning_knob")


def get_gemm_precision() -> str:
    return _precision


def precision_code() -> int:
    """The ``precision`` argument (TT_PRECISION_F32 / _BF16X3 / _BF16) of tt_linear_fwd, tt_label_propagate[_maps], tt_mlp_head_forward,
    tt_scores_sinkhorn and tt_vit_params.precision in the current mode."""
    return _PRECISIONS[_precision][0]


def plane_count() -> int:
    """bf16 planes the launch sequences use for blocks that keep no activations (0 = fp32 operands; 2 = fp16 PAIRS, see ``pairs()``)."""
    return _PRECISIONS[_precision][1]


def pairs() -> bool:
    """True in the "f16x3" mode: the blocks that keep nothing run on fp16-pair operands (``linear_fwd_pairs``)."""
    return _PRECISIONS[_precision][1] == 2


# The "f16x3" mode is a DISPATCH decision between two fp32-class arithmetics: below this many token rows per launch the pair kernels'
# extra passes (operand splits, the gradient's max pass) cost more than three-MFMA products save, so small launches keep the exact-f32
# kernels.  Measured on BASELINE C1 (4 frames: 788-row block launches, 394 / 392-row kept-frame and head launches), ms per step by
# threshold.  Round 4 (launch by launch): 1536 -> 3.10 (everything on f32), 700 -> 2.68, 512 -> 2.79, 393 -> 3.47, 300 -> 2.89, 200 -> 3.23:
# 640.  Round 5 (the step replayed as a HIP graph - launch counts no longer cost host time - and the general pair kernel on a ring of
# three / four K-tiles): 640 -> 2.07-2.11, 400 -> 2.08, 393 -> 2.18 (the kept frames' 394 rows on pairs, the head's 392 not: conversions),
# 300 -> 1.98-2.02, 200 -> 1.97-2.02, 0 -> 1.99-2.07: 256.  Tests set it to 0 to drive the pair kernels with tiny models.
PAIRS_MIN_ROWS = int(os.environ.get("TT_PAIRS_MIN_ROWS", "256"))   # (the environment variable: sweeps only)
GRAD_SCALE = os.environ.get("TT_NO_GRAD_SCALE") != "1"   # gradients are scaled by a power of two before their pair split (A/B aid: off)
AMAX_FROM_PRODUCERS = os.environ.get("TT_NO_AMAX_POOL") != "1"   # the gradient scale's maximum from the kernel that produced the dy (A/B aid: off = a max pass per dy)
TN_WGRAD = True   # weight gradients of the "f16x3" mode from row pairs (gemm_pairs_tn.hip); False: the transposed-operand route (A/B, tests)


def plane_count_for(rows: int) -> int:
    """``plane_count()`` for a launch sequence over ``rows`` token rows: the pair mode applies from PAIRS_MIN_ROWS rows on."""
    n = _PRECISIONS[_precision][1]
    return 0 if (n == 2 and rows < PAIRS_MIN_ROWS) else n


# Bumped by every op that rewrites parameters through raw pointers (AdamW, EMA, prototype renormalisation): tensors touched that way do not
# advance torch's version counter, so caches of derived operands (engine.weight_planes: the per-step pair split of a TRAINABLE weight)
# key on this as well.
PARAM_EPOCH = 0


def _bump_param_epoch() -> None:
    global PARAM_EPOCH
    PARAM_EPOCH += 1


# bench.py sets PROFILE to a list to get (layout, tile_choice, flops, start_event, end_event) per GEMM launch,
# recorded with HIP events on the stream the kernel is launched on.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, name: str, M: int, N: int, K: int, batch: int = 1):
    """name "NT" = a forward Linear: booked as "NT" only when the lean whole-tile kernel really ran it (tt_linear_fwd_route),
    as "NTgen" when the general kernel did, as "NTbf16" in the on-the-fly bf16 modes."""
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    lib = _lib.load()
    tile = lib.tt_gemm_tile_choice(M, N, batch)
    if name == "NT":
        if _PRECISIONS[_precision][0] != 0:
            name = "NTbf16"
        else:
            route = lib.tt_linear_fwd_route(M, N, K)
            tile, name = route & 3, ("NT" if route & 256 else "NTgen")
    PROFILE.append((name, tile, 2.0 * M * N * K * batch, e0, e1))


# ---- Linear ------------------------------------------------------------------------------------

def linear_fwd(x, w, bias=None, residual=None, act: int = 0, save_pre: bool = False, out=None):
    """y = act(x @ w.T + bias) (+ residual).  x [M,K], w [N,K].  Returns y or (y, pre_act)."""
    lib = _lib.load()
    _chk(x, "x"); _chk(w, "w")
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    y = out if out is not None else torch.empty((M, N), dtype=f32, device=x.device)
    pre = torch.empty((M, N), dtype=f32, device=x.device) if save_pre else None
    if bias is not None: _chk(bias, "bias")
    if residual is not None: _chk(residual, "residual")
    e0 = _prof_begin()
    _lib.check(lib.tt_linear_fwd(_p(x), _p(w), _p(bias), _p(residual), _p(y), _p(pre), M, N, K, act, precision_code(), _stream()), "tt_linear_fwd")
    _prof_end(e0, "NT", M, N, K)
    return (y, pre) if save_pre else y


def linear_bwd_data(dy, w, gelu_pre=None, out=None):
    """dx = dy @ w (* gelu'(gelu_pre))."""
    lib = _lib.load()
    _chk(dy, "dy"); _chk(w, "w")
    M, N = dy.shape
    K = w.shape[1]
    dx = out if out is not None else torch.empty((M, K), dtype=f32, device=dy.device)
    e0 = _prof_begin()
    _lib.check(lib.tt_linear_bwd_data(_p(dy), _p(w), _p(gelu_pre), _p(dx), M, N, K, _stream()), "tt_linear_bwd_data")
    _prof_end(e0, "NN", M, K, N)
    return dx


def linear_bwd_weight(dy, x, need_bias=True, dw_out=None, db_out=None):
    """dw = dy.T @ x, db = dy.sum(0)."""
    lib = _lib.load()
    _chk(dy, "dy"); _chk(x, "x")
    M, N = dy.shape
    K = x.shape[1]
    dw = dw_out if dw_out is not None else torch.empty((N, K), dtype=f32, device=dy.device)
    db = (db_out if db_out is not None else torch.empty((N,), dtype=f32, device=dy.device)) if need_bias else None
    nb = lib.tt_linear_bwd_weight_workspace_bytes(M, N, K)
    ws = _ws(nb, dy.device)
    if PROFILE is not None:  # time the GEMM alone: issue the bias column-sum as its own call
        e0 = _prof_begin()
        _lib.check(lib.tt_linear_bwd_weight(_p(dy), _p(x), _p(dw), None, M, N, K, _p(ws), nb, _stream()), "tt_linear_bwd_weight")
        _prof_end(e0, "TN", N, K, M)
        if db is not None:
            _lib.check(lib.tt_colsum(_p(dy), _p(db), M, N, _p(ws), nb, _stream()), "tt_colsum")
        return dw, db
    _lib.check(lib.tt_linear_bwd_weight(_p(dy), _p(x), _p(dw), _p(db), M, N, K, _p(ws), nb, _stream()), "tt_linear_bwd_weight")
    return dw, db


def linear_bwd(dy, w, x, gelu_pre=None, need_bias=True, dw_out=None, db_out=None):
    """Both backward products of an nn.Linear: dx = dy @ w (* gelu'(gelu_pre)), dw = dy.T @ x, db = dy.sum(0) - ONE launch for the
    two GEMMs where the lean kernels apply (tt_linear_bwd), bit-identical to linear_bwd_data + linear_bwd_weight.  ``dw_out`` / ``db_out``:
    caller-owned destinations (the data-parallel exchange's flat gradient buckets)."""
    lib = _lib.load()
    _chk(dy, "dy"); _chk(w, "w"); _chk(x, "x")
    M, N = dy.shape
    K = w.shape[1]
    assert w.shape[0] == N and tuple(x.shape) == (M, K), (dy.shape, w.shape, x.shape)
    if gelu_pre is not None: _chk(gelu_pre, "gelu_pre")
    dx = torch.empty((M, K), dtype=f32, device=dy.device)
    dw = _chk(dw_out, "dw_out") if dw_out is not None else torch.empty((N, K), dtype=f32, device=dy.device)
    db = (_chk(db_out, "db_out") if db_out is not None else torch.empty((N,), dtype=f32, device=dy.device)) if need_bias else None
    nb = lib.tt_linear_bwd_weight_workspace_bytes(M, N, K)
    ws = _ws(nb, dy.device)
    e0 = _prof_begin()   # booked as ONE entry: both products (4 M N K flops) and the split-K fold they end in
    _lib.check(lib.tt_linear_bwd(_p(dy), _p(w), _p(x), _p(gelu_pre), _p(dx), _p(dw), _p(db), M, N, K, _p(ws), nb, _stream()), "tt_linear_bwd")
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append(("BWD", 3, 4.0 * M * N * K, e0, e1))
    return dx, dw, db


def colsum(a):
    lib = _lib.load()
    _chk(a, "a")
    M, N = a.shape
    out = torch.empty((N,), dtype=f32, device=a.device)
    nb = lib.tt_colsum_workspace_bytes(M, N)
    ws = _ws(nb, a.device)
    _lib.check(lib.tt_colsum(_p(a), _p(out), M, N, _p(ws), nb, _stream()), "tt_colsum")
    return out


def gemm(A, B, a_mmajor=False, b_nmajor=False, alpha=1.0, out=None):
    """C = alpha * op(A) @ op(B) (see tt_gemm_f32).  2-D operands, or 3-D for a batch."""
    lib = _lib.load()
    _chk(A, "A"); _chk(B, "B")
    batch = 1
    sA = sB = sC = 0
    if A.dim() == 3:
        batch = A.shape[0]
        A2, B2 = A[0], B[0]
        sA, sB = A.stride(0), B.stride(0)
    else:
        A2, B2 = A, B
    if a_mmajor:
        K, M = A2.shape
    else:
        M, K = A2.shape
    if b_nmajor:
        Kb, N = B2.shape
    else:
        N, Kb = B2.shape
    assert K == Kb, (A.shape, B.shape)
    shape = (batch, M, N) if A.dim() == 3 else (M, N)
    Cm = out if out is not None else torch.empty(shape, dtype=f32, device=A.device)
    sC = M * N if batch > 1 else 0
    _lib.check(lib.tt_gemm_f32(_p(A), _p(B), _p(Cm), M, N, K, A2.stride(0), B2.stride(0), N, int(a_mmajor), int(b_nmajor),
                               float(alpha), batch, sA, sB, sC, _stream()), "tt_gemm_f32")
    return Cm


# ---- ViT pieces ----------------------------------------------------------------------------------

def patch_embed_fwd(img, w, bias, cls, pos, patch: int, frame_map=None):
    """img [F_src,C,H,W] -> tokens [F, n+1, D] (conv k=s=patch, cls token, pos-embed)."""
    lib = _lib.load()
    _chk(img, "img"); _chk(w, "w"); _chk(bias, "bias"); _chk(cls, "cls"); _chk(pos, "pos")
    Fs, Cc, H, W = img.shape
    D = w.shape[0]
    F = Fs if frame_map is None else frame_map.numel()
    if frame_map is not None: _chk(frame_map, "frame_map", torch.int32)
    n = (H // patch) * (W // patch)
    if pos.numel() != (n + 1) * D:
        raise ValueError("pos_embed does not match the token grid: pass VisionTransformer.pos_table(H, W)")
    tokens = torch.empty((F, n + 1, D), dtype=f32, device=img.device)
    e0 = _prof_begin()
    _lib.check(lib.tt_patch_embed_fwd(_p(img), _p(frame_map), _p(w), _p(bias), _p(cls), _p(pos), _p(tokens), F, Cc, H, W, patch, D,
                                      _stream()), "tt_patch_embed_fwd")
    _prof_end(e0, "patch", F * n, D, Cc * patch * patch)
    return tokens


def patch_embed_planes_ok(patch: int, W: int, K: int, D: int) -> bool:
    """Shapes ``patch_embed_fwd_planes`` takes (tt_patch_embed_fwd_planes)."""
    return patch % 4 == 0 and W % 4 == 0 and K % 64 == 0 and D % 64 == 0


def patch_embed_pairs_ok(patch: int, W: int, K: int, D: int) -> bool:
    """Shapes ``patch_embed_fwd_pairs`` takes (tt_patch_embed_fwd_pairs)."""
    return patch % 4 == 0 and W % 4 == 0 and K % 32 == 0 and D % 64 == 0


def patch_embed_fwd_pairs(img, w_pairs, bias, cls, pos, patch: int, frame_map=None):
    """``patch_embed_fwd`` on fp16-pair operands (the "f16x3" mode): w_pairs [D, 2 C P P] = ``split_pairs`` of the conv weight viewed
    [D, C P P]; the patches are split into pairs on their way into an im2col buffer, ONE pair GEMM over all token rows."""
    lib = _lib.load()
    _chk(img, "img"); _chk(w_pairs, "w_pairs", f16); _chk(bias, "bias"); _chk(cls, "cls"); _chk(pos, "pos")
    Fs, Cc, H, W = img.shape
    D = w_pairs.shape[0]
    F = Fs if frame_map is None else frame_map.numel()
    if frame_map is not None: _chk(frame_map, "frame_map", torch.int32)
    n = (H // patch) * (W // patch)
    K = Cc * patch * patch
    assert w_pairs.shape == (D, 2 * K), (w_pairs.shape, D, K)
    if pos.numel() != (n + 1) * D:
        raise ValueError("pos_embed does not match the token grid: pass VisionTransformer.pos_table(H, W)")
    tokens = torch.empty((F, n + 1, D), dtype=f32, device=img.device)
    nb = lib.tt_patch_embed_pairs_workspace_bytes(F, Cc, H, W, patch)
    ws = _ws(nb, img.device)
    M = F * (n + 1)
    p8 = PROFILE is not None and lib.tt_linear_fwd_pairs_route(M, D, K, 0, 1, 1, 1, 0, 0) == 8
    e0 = _prof_begin()
    _lib.check(lib.tt_patch_embed_fwd_pairs(_p(img), _p(frame_map), _p(w_pairs), _p(bias), _p(cls), _p(pos), _p(tokens), F, Cc, H, W, patch, D,
                                            _p(ws), nb, _p(range_flag(img.device)), _stream()), "tt_patch_embed_fwd_pairs")
    _prof_end(e0, "PAIRS8" if p8 else "PAIRS", M, D, K)
    return tokens


def patch_embed_fwd_planes(img, w_planes, bias, cls, pos, patch: int, frame_map=None):
    """``patch_embed_fwd`` on bf16 operands (the "bf16" precision mode, BASELINE C4's path): w_planes [1, D, C*P*P] bf16 =
    ``split_planes(w, 1)``; the patches are rounded to bf16 on their way into an im2col buffer, fp32 accumulation, fp32 tokens."""
    lib = _lib.load()
    _chk(img, "img"); _chk(w_planes, "w_planes", bf16); _chk(bias, "bias"); _chk(cls, "cls"); _chk(pos, "pos")
    Fs, Cc, H, W = img.shape
    D = w_planes.shape[1]
    F = Fs if frame_map is None else frame_map.numel()
    if frame_map is not None: _chk(frame_map, "frame_map", torch.int32)
    n = (H // patch) * (W // patch)
    K = Cc * patch * patch
    if pos.numel() != (n + 1) * D:
        raise ValueError("pos_embed does not match the token grid: pass VisionTransformer.pos_table(H, W)")
    tokens = torch.empty((F, n + 1, D), dtype=f32, device=img.device)
    nb = lib.tt_patch_embed_planes_workspace_bytes(F, Cc, H, W, patch)
    ws = _ws(nb, img.device)
    M = F * (n + 1)
    p8 = bool(lib.tt_linear_fwd_planes_route(1, M, D, K, 0, 1, 1, 1, 0, 0))
    e0 = _prof_begin()
    _lib.check(lib.tt_patch_embed_fwd_planes(_p(img), _p(frame_map), _p(w_planes), _p(bias), _p(cls), _p(pos), _p(tokens), F, Cc, H, W, patch, D,
                                             _p(ws), nb, _stream()), "tt_patch_embed_fwd_planes")
    _prof_end(e0, "PLANES8_1" if p8 else "PLANES1", M, D, K)
    return tokens


def pos_embed_interpolate(pos, grid_h: int, grid_w: int):
    """interpolate_pos_encoding's bicubic branch (dino_vision_transformer.py:219-234): pos [1+g*g, D] -> [1+grid_h*grid_w, D]."""
    lib = _lib.load()
    _chk(pos, "pos")
    D = pos.shape[-1]
    g = int(round((pos.shape[0] - 1) ** 0.5))
    out = torch.empty((1 + grid_h * grid_w, D), dtype=f32, device=pos.device)
    # the reference adds 0.1 to the target grid before dividing ("to avoid floating point error in the interpolation")
    _lib.check(lib.tt_pos_embed_interpolate(_p(pos), _p(out), g, grid_h, grid_w, D, (grid_h + 0.1) / g, (grid_w + 0.1) / g, _stream()),
               "tt_pos_embed_interpolate")
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-6, save_stats=False, out=None, drop_first_token=False):
    """LayerNorm over the last dim.  drop_first_token: x is [F,N,D] and the result is [F*(N-1), D] (cls row removed)."""
    lib = _lib.load()
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta")
    D = x.shape[-1]
    rows = x.numel() // D
    skip = 0
    if drop_first_token:
        skip = x.shape[-2]
        rows = rows // skip * (skip - 1)
        y = out if out is not None else torch.empty((rows, D), dtype=f32, device=x.device)
    else:
        y = out if out is not None else torch.empty_like(x)
    mean = torch.empty((rows,), dtype=f32, device=x.device) if save_stats else None
    rstd = torch.empty((rows,), dtype=f32, device=x.device) if save_stats else None
    _lib.check(lib.tt_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, float(eps), skip, _stream()), "tt_layernorm_fwd")
    return (y, mean, rstd) if save_stats else y


def layernorm_bwd(dy, x, gamma, mean, rstd, need_wgrad=True, dx_accum=None, drop_first_token=False, dg_out=None, db_out=None, amax_out=None):
    """Returns (dx, dgamma, dbeta).  dx_accum: tensor to accumulate dx into (residual branch).
    drop_first_token: dy is [F*(N-1), D] against x [F,N,D]; dx is [F,N,D] with zero cls rows.
    ``amax_out``: a zeroed 1-float device tensor (``AmaxPool.take``) the kernel raises to max |dx| - dx is the next Linear's dy, whose pair
    split then needs no max pass (``linear_bwd_pairs(dy_amax=)``)."""
    lib = _lib.load()
    _chk(dy, "dy"); _chk(x, "x")
    D = x.shape[-1]
    rows = dy.numel() // D
    skip = x.shape[-2] if drop_first_token else 0
    if dx_accum is not None:
        dx = dx_accum
    elif drop_first_token:
        dx = torch.zeros_like(x)
    else:
        dx = torch.empty_like(x)
    dg = (_chk(dg_out, "dg_out") if dg_out is not None else torch.empty((D,), dtype=f32, device=x.device)) if need_wgrad else None
    db = (_chk(db_out, "db_out") if db_out is not None else torch.empty((D,), dtype=f32, device=x.device)) if need_wgrad else None
    nb = lib.tt_layernorm_bwd_workspace_bytes(rows, D)
    ws = _ws(nb, x.device)
    _lib.check(lib.tt_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), rows, D,
                                    1 if dx_accum is not None else 0, skip, _p(ws), nb, _p(amax_out), _stream()), "tt_layernorm_bwd")
    return dx, dg, db


def attention_fwd(qkv, num_heads: int, save_lse=False, return_probs=False):
    """qkv [F,N,3*D] -> out [F,N,D] (+ lse [F,H,N], probs [F,H,N,N])."""
    lib = _lib.load()
    _chk(qkv, "qkv")
    F, N, D3 = qkv.shape
    D = D3 // 3
    hd = D // num_heads
    out = torch.empty((F, N, D), dtype=f32, device=qkv.device)
    lse = torch.empty((F, num_heads, N), dtype=f32, device=qkv.device) if save_lse else None
    probs = torch.empty((F, num_heads, N, N), dtype=f32, device=qkv.device) if return_probs else None
    _lib.check(lib.tt_attention_fwd(_p(qkv), _p(out), _p(lse), _p(probs), F, N, num_heads, hd, float(hd ** -0.5), _stream()),
               "tt_attention_fwd")
    return out, lse, probs


# the "f16x3" mode's attention backward with its row-wise products on fp16 pairs (tt_attention_bwd_pairs); "0": the fp32-MFMA kernels (A/B)
ATTN_BWD_PAIRS = os.environ.get("TT_ATTN_BWD_PAIRS", "1") != "0"


def attention_bwd(qkv, out, dout, lse, num_heads: int, bf16_products: bool = False, amax_out=None, pair_products: bool = False, dout_amax=None):
    """dqkv of the fused attention core from the forward's out / lse.  ``bf16_products``: the matrix products on bf16 MFMA
    (tt_attention_bwd_bf16 - the "bf16" precision mode's backward; statistics, P and dS stay fp32); ``pair_products``: S and dP on fp16
    pairs (tt_attention_bwd_pairs - the "f16x3" mode's: fp32-class; the range flag is the pair entry points'; ``dout_amax``: the amax
    slot the kernel that wrote dout raised - without one the call measures max |dout| itself)."""
    lib = _lib.load()
    _chk(qkv, "qkv"); _chk(out, "out"); _chk(dout, "dout"); _chk(lse, "lse")
    F, N, D3 = qkv.shape
    hd = D3 // 3 // num_heads
    dqkv = torch.empty_like(qkv)
    if pair_products and not bf16_products:
        nb = lib.tt_attention_bwd_pairs_workspace_bytes(F, N, num_heads, hd)
        ws = _ws(nb, qkv.device)
        _lib.check(lib.tt_attention_bwd_pairs(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), F, N, num_heads, hd, float(hd ** -0.5), _p(dout_amax), _p(ws), nb,
                                              _p(range_flag(qkv.device)), _p(amax_out), _stream()), "tt_attention_bwd_pairs")
        return dqkv
    nb = lib.tt_attention_bwd_workspace_bytes(F, N, num_heads, hd)
    ws = _ws(nb, qkv.device)
    if bf16_products:
        _lib.check(lib.tt_attention_bwd_bf16(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), F, N, num_heads, hd, float(hd ** -0.5), _p(ws), nb, _stream()),
                   "tt_attention_bwd_bf16")
    else:   # (amax_out: max |dqkv| for the pair split of the qkv Linear's dy, as layernorm_bwd)
        _lib.check(lib.tt_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), F, N, num_heads, hd, float(hd ** -0.5), _p(ws), nb, _p(amax_out),
                                        _stream()), "tt_attention_bwd")
    return dqkv


# ---- prototypes / assignment ---------------------------------------------------------------------

def l2norm_fwd(x, save_inv=False, out=None):
    """F.normalize(x, dim=-1).  x may be a row-strided 2-D view (stride(1) == 1)."""
    lib = _lib.load()
    if x.dim() != 2 or x.stride(1) != 1 or not x.is_cuda or x.dtype != f32:
        raise ValueError("l2norm_fwd: expected a 2-D fp32 GPU tensor with unit inner stride")
    rows, D = x.shape
    xn = out if out is not None else torch.empty((rows, D), dtype=f32, device=x.device)
    inv = torch.empty((rows,), dtype=f32, device=x.device) if save_inv else None
    _lib.check(lib.tt_l2norm_fwd(_p(x), x.stride(0), _p(xn), _p(inv), rows, D, _stream()), "tt_l2norm_fwd")
    return (xn, inv) if save_inv else xn


def l2norm_bwd(dxn, xn, inv, amax_out=None):
    lib = _lib.load()
    _chk(dxn, "dxn"); _chk(xn, "xn"); _chk(inv, "inv")
    rows, D = xn.shape
    dx = torch.empty_like(xn)
    _lib.check(lib.tt_l2norm_bwd(_p(dxn), _p(xn), _p(inv), _p(dx), rows, D, _p(amax_out), _stream()), "tt_l2norm_bwd")
    return dx


class AmaxPool:
    """Zeroed 1-float device slots for the maxima the gradient PRODUCERS publish (``amax_out`` of layernorm_bwd / attention_bwd /
    l2norm_bwd / the gelu' data gradient) and the pair split of that gradient consumes (``linear_bwd_pairs(dy_amax=)``): one ``reset()`` (a
    64-float fill) at the top of a backward instead of a max pass over every dy (12 per C2 step, 0.11 ms)."""
    _pools: dict = {}

    def __init__(self, device):
        self.SLOT = _lib.load().tt_amax_slot_bytes() // 4   # floats per slot: 16 ways, 64 floats apart (the producers spread their atomics)
        self.buf = torch.zeros(24 * self.SLOT, dtype=f32, device=device)
        self.n = 0

    @classmethod
    def get(cls, device) -> "AmaxPool":
        dev = torch.device(device)
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
        p = cls._pools.get(key)
        if p is None:
            p = cls._pools[key] = cls(dev)
        return p

    def reset(self) -> None:
        self.buf.zero_()
        self.n = 0

    def take(self):
        if (self.n + 1) * self.SLOT > self.buf.numel():
            return None            # (more gradients than slots: those splits make their own max pass)
        self.n += 1
        return self.buf[(self.n - 1) * self.SLOT:self.n * self.SLOT]


def normalize_rows_(w):
    _bump_param_epoch()
    lib = _lib.load()
    _chk(w, "w")
    _lib.check(lib.tt_normalize_rows_inplace(_p(w), w.shape[0], w.shape[1], _stream()), "tt_normalize_rows_inplace")
    return w


def sinkhorn(scores, iters: int, eps: float = 0.05, row0: int = 0, rows_out: Optional[int] = None):
    """scores [B_total, K] -> q [rows_out, K] for rows [row0, row0+rows_out)."""
    lib = _lib.load()
    _chk(scores, "scores")
    B, K = scores.shape
    rows_out = B - row0 if rows_out is None else rows_out
    q = torch.empty((rows_out, K), dtype=f32, device=scores.device)
    nb = lib.tt_sinkhorn_workspace_bytes(B, K)
    ws = _ws(nb, scores.device)
    _lib.check(lib.tt_sinkhorn(_p(scores), _p(q), B, K, row0, rows_out, float(eps), int(iters), _p(ws), nb, _stream()), "tt_sinkhorn")
    return q


class SinkhornLocal:
    """One rank's share of a DISTRIBUTED solve in the reference's own form (my_utils.py:250-272): the columns stay on their rank, the K
    row sums are all-reduced once per iteration by the caller - ``u = begin(scores)``; ``iters - 1`` times ``u = step(all_reduce(u))``;
    ``q = end(all_reduce(u))`` (``end(None)`` for zero iterations).  ``lib``: the HIP library (default) or the CPU twin (tests)."""

    def __init__(self, B_loc: int, B_total: int, K: int, device, lib=None):
        self.lib = _lib.load() if lib is None else lib
        self.pre = "tt_" if lib is None else "tt_cpu_"
        self.B_loc, self.B_total, self.K = int(B_loc), int(B_total), int(K)
        self.nb = getattr(self.lib, self.pre + "sinkhorn_local_workspace_bytes")(self.B_loc, self.K)
        self.ws = torch.empty((self.nb + 3) // 4, dtype=f32, device=device)   # owned: it holds E between the calls
        self.device = device

    def _stream(self):
        return _stream() if torch.device(self.device).type == "cuda" else None

    def _ck(self, rc: int, what: str) -> None:
        if rc != 0:
            if self.pre == "tt_":
                _lib.check(rc, "tt_" + what)
            raise RuntimeError(f"{self.pre}{what} failed (code {rc})")

    def begin(self, scores, eps: float):
        assert scores.shape == (self.B_loc, self.K) and scores.dtype == f32 and scores.is_contiguous()
        u = torch.empty((self.K,), dtype=f32, device=scores.device)
        self._ck(getattr(self.lib, self.pre + "sinkhorn_local_begin")(_p(scores), _p(u), self.B_loc, self.K, float(eps), _p(self.ws), self.nb, self._stream()),
                 "sinkhorn_local_begin")
        return u

    def step(self, u_global):
        u = torch.empty_like(u_global)
        self._ck(getattr(self.lib, self.pre + "sinkhorn_local_step")(_p(u_global), _p(u), self.B_loc, self.B_total, self.K, _p(self.ws), self.nb, self._stream()),
                 "sinkhorn_local_step")
        return u

    def end(self, u_global, rows_out: Optional[int] = None):
        rows_out = self.B_loc if rows_out is None else int(rows_out)
        q = torch.empty((rows_out, self.K), dtype=f32, device=self.ws.device)
        self._ck(getattr(self.lib, self.pre + "sinkhorn_local_end")(_p(u_global), _p(q), self.B_loc, rows_out, self.K, _p(self.ws), self.nb, self._stream()),
                 "sinkhorn_local_end")
        return q


def sinkhorn_from_q(Q, iters: int, row0: int = 0, rows_out: Optional[int] = None, transposed: bool = False):
    """Q positive = exp(scores / eps): [K, B_total] as my_utils.sinkhorn receives it, or [B_total, K] with ``transposed``
    -> q [rows_out, K]."""
    lib = _lib.load()
    _chk(Q, "Q")
    B, K = Q.shape if transposed else Q.shape[::-1]
    rows_out = B - row0 if rows_out is None else rows_out
    q = torch.empty((rows_out, K), dtype=f32, device=Q.device)
    nb = lib.tt_sinkhorn_workspace_bytes(B, K)
    ws = _ws(nb, Q.device)
    _lib.check(lib.tt_sinkhorn_from_q(_p(Q), int(transposed), _p(q), B, K, row0, rows_out, int(iters), _p(ws), nb, _stream()),
               "tt_sinkhorn_from_q")
    return q


def label_propagate_sims(xn, K: int, n_last_frames=7):
    """The first half of ``label_propagate`` - the cosine similarities of all target frames, which do not depend on the assignment - for a
    caller that computes them beside other work (``label_propagate(..., sims=)`` takes the result).  None when they do not fit one chunk."""
    lib = _lib.load()
    _chk(xn, "xn")
    fs, bs, n, D = xn.shape
    g = int(round(n ** 0.5))
    assert g * g == n
    nb = lib.tt_label_propagate_workspace_bytes(bs, fs, g, D, K, n_last_frames)
    ws = _ws(nb, xn.device)
    rc = lib.tt_label_propagate_sims(_p(xn), bs, fs, g, D, K, n_last_frames, precision_code(), _p(ws), nb, _stream())
    if rc == 1:
        return None
    _lib.check(rc, "tt_label_propagate_sims")
    return ws


def label_propagate(xn, seg0, n_last_frames=7, radius=6, topk=5, temperature=0.1, return_pmap=False, sims=None):
    """xn [fs,bs,n,D] normalised tokens (time-major), seg0 [bs,n,K] -> labels [bs,n] int64 (+ pmap [bs,n,K] fp64).  ``sims``: the workspace
    ``label_propagate_sims`` prepared for the same xn / K / n_last_frames."""
    lib = _lib.load()
    _chk(xn, "xn"); _chk(seg0, "seg0")
    fs, bs, n, D = xn.shape
    K = seg0.shape[-1]
    g = int(round(n ** 0.5))
    assert g * g == n
    labels = torch.empty((bs, n), dtype=torch.int64, device=xn.device)
    pmap = torch.empty((bs, n, K), dtype=torch.float64, device=xn.device) if return_pmap else None
    nb = lib.tt_label_propagate_workspace_bytes(bs, fs, g, D, K, n_last_frames)
    if sims is not None:
        assert sims.numel() >= nb, (sims.numel(), nb)
        _lib.check(lib.tt_label_propagate_from_sims(_p(xn), _p(seg0), _p(labels), _p(pmap), bs, fs, g, D, K, n_last_frames, radius, topk,
                                                    float(temperature), _p(sims), sims.numel(), _stream()), "tt_label_propagate_from_sims")
        return (labels, pmap) if return_pmap else labels
    ws = _ws(nb, xn.device)
    _lib.check(lib.tt_label_propagate(_p(xn), _p(seg0), _p(labels), _p(pmap), bs, fs, g, D, K, n_last_frames, radius, topk,
                                      float(temperature), precision_code(), _p(ws), nb, _stream()), "tt_label_propagate")
    return (labels, pmap) if return_pmap else labels


def label_propagate_maps(xn, seg0, n_last_frames=7, radius=6, topk=5, temperature=0.1):
    """As ``label_propagate`` but returns ALL propagated maps [fs-1, bs, n, K] fp64 (mask_propagation.py:448-496)."""
    lib = _lib.load()
    _chk(xn, "xn"); _chk(seg0, "seg0")
    fs, bs, n, D = xn.shape
    K = seg0.shape[-1]
    g = int(round(n ** 0.5))
    assert g * g == n
    maps = torch.empty((fs - 1, bs, n, K), dtype=torch.float64, device=xn.device)
    nb = lib.tt_label_propagate_workspace_bytes(bs, fs, g, D, K, n_last_frames)
    ws = _ws(nb, xn.device)
    _lib.check(lib.tt_label_propagate_maps(_p(xn), _p(seg0), _p(maps), bs, fs, g, D, K, n_last_frames, radius, topk, float(temperature),
                                           precision_code(), _p(ws), nb, _stream()), "tt_label_propagate_maps")
    return maps


def upsample_argmax(maps, resolution: int):
    """maps [M, n, K] fp64 -> labels [M, R, R] int64 = argmax_K of the bilinear (align_corners=False) upsampling."""
    lib = _lib.load()
    _chk(maps, "maps", torch.float64)
    M, n, K = maps.shape
    g = int(round(n ** 0.5))
    assert g * g == n
    out = torch.empty((M, resolution, resolution), dtype=torch.int64, device=maps.device)
    _lib.check(lib.tt_upsample_argmax(_p(maps), _p(out), M, g, K, int(resolution), _stream()), "tt_upsample_argmax")
    return out


def confusion_counts(pred, gt, num_classes: int):
    """counts[gt, pred] over all elements, int64 [C, C] (labels outside [0, C) are ignored)."""
    lib = _lib.load()
    _chk(pred, "pred", torch.int64); _chk(gt, "gt", torch.int64)
    if pred.numel() != gt.numel():
        raise ValueError("pred and gt need the same number of elements")
    counts = torch.empty((num_classes, num_classes), dtype=torch.int64, device=pred.device)
    _lib.check(lib.tt_confusion_counts(_p(pred), _p(gt), pred.numel(), int(num_classes), _p(counts), _stream()), "tt_confusion_counts")
    return counts


# ---- clip input pipeline (uint8 frames [F, H, W, 3]) ------------------------------------------------------------------

def img_resample_h(frames, coeffs, bounds, y0: int, x0: int, h: int):
    """Horizontal pass of Pillow's bilinear resize over rows y0..y0+h, columns from x0: -> uint8 [F, h, OW, 3]."""
    lib = _lib.load()
    _chk(frames, "frames", torch.uint8); _chk(coeffs, "coeffs", torch.int32); _chk(bounds, "bounds", torch.int32)
    Fr, H, W, _ = frames.shape
    OW, ksize = coeffs.shape
    out = torch.empty((Fr, h, OW, 3), dtype=torch.uint8, device=frames.device)
    _lib.check(lib.tt_img_resample_h(_p(frames), _p(out), _p(coeffs), _p(bounds), Fr, H, W, y0, x0, h, OW, ksize, _stream()), "tt_img_resample_h")
    return out


def img_resample_v(frames, coeffs, bounds, y0: int = 0, to_tensor=None, flip: bool = False):
    """Vertical pass: -> uint8 [F, OH, W, 3], or with ``to_tensor=(mean, std)`` the finished float32 [F, 3, OH, W]."""
    import ctypes as C

    lib = _lib.load()
    _chk(frames, "frames", torch.uint8); _chk(coeffs, "coeffs", torch.int32); _chk(bounds, "bounds", torch.int32)
    Fr, Hin, W, _ = frames.shape
    OH, ksize = coeffs.shape
    if to_tensor is None:
        out = torch.empty((Fr, OH, W, 3), dtype=torch.uint8, device=frames.device)
        _lib.check(lib.tt_img_resample_v(_p(frames), _p(out), None, _p(coeffs), _p(bounds), Fr, Hin, W, y0, OH, ksize, 0, None, None, _stream()),
                   "tt_img_resample_v")
        return out
    mean, std = to_tensor
    out = torch.empty((Fr, 3, OH, W), dtype=f32, device=frames.device)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(lib.tt_img_resample_v(_p(frames), None, _p(out), _p(coeffs), _p(bounds), Fr, Hin, W, y0, OH, ksize, int(bool(flip)), m3, s3,
                                     _stream()), "tt_img_resample_v")
    return out


IMG_GRAYSCALE, IMG_BRIGHTNESS, IMG_CONTRAST, IMG_SATURATION, IMG_HUE = range(5)


def img_color_(frames, mode: int, factor: float = 1.0, hue_shift: int = 0):
    """In-place colour operation on uint8 [F, H, W, 3] (modes above)."""
    lib = _lib.load()
    _chk(frames, "frames", torch.uint8)
    Fr, H, W, _ = frames.shape
    ws = torch.empty(Fr, dtype=torch.int64, device=frames.device) if mode == IMG_CONTRAST else None
    _lib.check(lib.tt_img_color(_p(frames), Fr, H, W, int(mode), float(factor), int(hue_shift), _p(ws), _stream()), "tt_img_color")
    return frames


def img_box_blur(frames, direction: int, radius: int, ww: int, fw: int):
    lib = _lib.load()
    _chk(frames, "frames", torch.uint8)
    Fr, H, W, _ = frames.shape
    out = torch.empty_like(frames)
    _lib.check(lib.tt_img_box_blur(_p(frames), _p(out), Fr, H, W, int(direction), int(radius), int(ww), int(fw), _stream()), "tt_img_box_blur")
    return out


def col_moments(x):
    """Per-column mean and population variance of x [rows, cols] (fp64 tensors)."""
    lib = _lib.load()
    _chk(x, "x")
    rows, cols = x.shape
    mean = torch.empty(cols, dtype=torch.float64, device=x.device)
    var = torch.empty_like(mean)
    nb = lib.tt_col_moments_workspace_bytes(rows, cols)
    ws = _ws(nb, x.device)
    _lib.check(lib.tt_col_moments(_p(x), _p(mean), _p(var), rows, cols, _p(ws), nb, _stream()), "tt_col_moments")
    return mean, var


def affine_cols_(x, scale, shift):
    """x[r, c] = x[r, c] * scale[c] + shift[c] in place."""
    lib = _lib.load()
    _chk(x, "x"); _chk(scale, "scale"); _chk(shift, "shift")
    rows, cols = x.shape
    _lib.check(lib.tt_affine_cols_inplace(_p(x), _p(scale), _p(shift), rows, cols, _stream()), "tt_affine_cols_inplace")
    return x


def upsample_bilinear_tokens(x, resolution: int):
    """x [M, g*g, C] fp32 -> [M, R*R, C] fp32 (bilinear, align_corners=False, fp64 arithmetic)."""
    lib = _lib.load()
    _chk(x, "x")
    M, n, Cc = x.shape
    g = int(round(n ** 0.5))
    assert g * g == n
    out = torch.empty((M, resolution * resolution, Cc), dtype=f32, device=x.device)
    _lib.check(lib.tt_upsample_bilinear_tokens(_p(x), _p(out), M, g, Cc, int(resolution), _stream()), "tt_upsample_bilinear_tokens")
    return out


def upsample_argmax_f32(maps, resolution: int):
    """maps [M, n, K] fp32 -> labels [M, R, R] int64 (fp32 bilinear interpolation, first maximum)."""
    lib = _lib.load()
    _chk(maps, "maps")
    M, n, K = maps.shape
    g = int(round(n ** 0.5))
    assert g * g == n
    out = torch.empty((M, resolution, resolution), dtype=torch.int64, device=maps.device)
    _lib.check(lib.tt_upsample_argmax_f32(_p(maps), _p(out), M, g, K, int(resolution), _stream()), "tt_upsample_argmax_f32")
    return out


def kmeans_assign(x, centroids, return_dist=False):
    """x [P, d], centroids [k, d] -> labels int32 [P] (+ squared distances)."""
    lib = _lib.load()
    _chk(x, "x"); _chk(centroids, "centroids")
    P, d = x.shape
    k = centroids.shape[0]
    labels = torch.empty(P, dtype=torch.int32, device=x.device)
    dist2 = torch.empty(P, dtype=f32, device=x.device) if return_dist else None
    _lib.check(lib.tt_kmeans_assign(_p(x), _p(centroids), _p(labels), _p(dist2), P, d, k, _stream()), "tt_kmeans_assign")
    return (labels, dist2) if return_dist else labels


def kmeans_accumulate(x, labels, k: int):
    """Sums [k, d] (fp64) and counts [k] (int64) of the points of each label."""
    lib = _lib.load()
    _chk(x, "x"); _chk(labels, "labels", torch.int32)
    P, d = x.shape
    sums = torch.empty((k, d), dtype=torch.float64, device=x.device)
    counts = torch.empty(k, dtype=torch.int64, device=x.device)
    nb = lib.tt_kmeans_accumulate_workspace_bytes(P, d, k)
    ws = _ws(nb, x.device)
    _lib.check(lib.tt_kmeans_accumulate(_p(x), _p(labels), _p(sums), _p(counts), P, d, k, _p(ws), nb, _stream()), "tt_kmeans_accumulate")
    return sums, counts


def ce_loss_fwd_bwd(scores, labels, temperature=0.1, need_grad=True, row_weight=None):
    """mean CE of scores/temperature vs labels (per-row weights = the --use_mask loss mask); returns (loss[1], dscores or None)."""
    lib = _lib.load()
    _chk(scores, "scores"); _chk(labels, "labels", torch.int64)
    if row_weight is not None:
        _chk(row_weight, "row_weight")
        if row_weight.numel() != scores.shape[0]:
            raise ValueError("row_weight needs one entry per score row")
    rows, K = scores.shape
    loss = torch.empty((1,), dtype=f32, device=scores.device)
    ds = torch.empty_like(scores) if need_grad else None
    nb = lib.tt_ce_workspace_bytes(rows)
    ws = _ws(nb, scores.device)
    _lib.check(lib.tt_ce_loss_fwd_bwd(_p(scores), _p(labels), _p(row_weight), _p(loss), _p(ds), rows, K, float(temperature), _p(ws), nb,
                                      _stream()), "tt_ce_loss_fwd_bwd")
    return loss, ds


def foreground_mask(qkv, num_heads: int, spatial_res: int, threshold: float = 0.65, blur_sigma: float = 0.6, kernel_size: int = 7,
                    return_aux: bool = False):
    """process_attentions (models.py:93-131) from the last block's qkv activations [F, N, 3*D] -> mask [F, g*g] in {0,1}
    (and, with ``return_aux``, the blurred head-mean attention and each pixel's distance to the mass cut)."""
    lib = _lib.load()
    _chk(qkv, "qkv")
    Fr, N, D3 = qkv.shape
    hd = D3 // 3 // num_heads
    n = N - 1
    mask = torch.empty((Fr, n), dtype=f32, device=qkv.device)
    blurred = torch.empty_like(mask) if return_aux else None
    margin = torch.empty_like(mask) if return_aux else None
    _lib.check(lib.tt_foreground_mask(_p(qkv), _p(mask), _p(blurred), _p(margin), Fr, N, num_heads, hd, spatial_res, float(hd) ** -0.5,
                                      float(threshold), float(blur_sigma), int(kernel_size), _stream()), "tt_foreground_mask")
    return (mask, blurred, margin) if return_aux else mask


def foreground_mask_from_probs(cls_probs, spatial_res: int, threshold: float = 0.65, blur_sigma: float = 0.6, kernel_size: int = 7,
                               return_aux: bool = False):
    """Same from row 0 of the attention probabilities, cls_probs [F, H, N]."""
    lib = _lib.load()
    _chk(cls_probs, "cls_probs")
    Fr, H, N = cls_probs.shape
    mask = torch.empty((Fr, N - 1), dtype=f32, device=cls_probs.device)
    blurred = torch.empty_like(mask) if return_aux else None
    margin = torch.empty_like(mask) if return_aux else None
    _lib.check(lib.tt_foreground_mask_from_probs(_p(cls_probs), _p(mask), _p(blurred), _p(margin), Fr, N, H, spatial_res, float(threshold),
                                                 float(blur_sigma), int(kernel_size), _stream()), "tt_foreground_mask_from_probs")
    return (mask, blurred, margin) if return_aux else mask


def scale_rows_(x, row_scale):
    """x[r, :] *= row_scale[r] in place (features * mask, models.py:142)."""
    lib = _lib.load()
    _chk(x, "x"); _chk(row_scale, "row_scale")
    rows, cols = x.shape
    if row_scale.numel() != rows:
        raise ValueError("row_scale needs one entry per row")
    _lib.check(lib.tt_scale_rows_inplace(_p(x), _p(row_scale), rows, cols, _stream()), "tt_scale_rows_inplace")
    return x


def queue_push_(queue, feats, idx):
    lib = _lib.load()
    _chk(queue, "queue"); _chk(feats, "feats"); _chk(idx, "idx", torch.int64)
    Q, D = queue.shape
    m = idx.numel()
    scratch = torch.empty_like(queue)
    _lib.check(lib.tt_queue_push(_p(queue), _p(scratch), _p(feats), _p(idx), Q, D, m, _stream()), "tt_queue_push")
    return queue


# ---- optimiser -------------------------------------------------------------------------------------

def adamw_step_(entries: Sequence[tuple], step: int, beta1=0.9, beta2=0.999, eps=1e-8):
    """entries: (param, grad, exp_avg, exp_avg_sq, lr, weight_decay) tuples, all fp32 GPU contiguous."""
    _bump_param_epoch()
    lib = _lib.load()
    cap = 40
    for i in range(0, len(entries), cap):
        chunk = entries[i:i + cap]
        arr = (_lib.AdamwTensor * len(chunk))()
        for j, (p, g, m, v, lr, wd) in enumerate(chunk):
            for t, nm in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
                _chk(t, nm)
            arr[j] = _lib.AdamwTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(wd))
        _lib.check(lib.tt_adamw_step(arr, len(chunk), int(step), float(beta1), float(beta2), float(eps), _stream()), "tt_adamw_step")


def scale_tensors_(tensors: Sequence[torch.Tensor], scale: torch.Tensor):
    """t *= scale for every tensor (scale: a one-element fp32 GPU tensor), one launch per 40 tensors."""
    lib = _lib.load()
    _chk(scale, "scale")
    cap = 40
    for i in range(0, len(tensors), cap):
        chunk = tensors[i:i + cap]
        arr = (_lib.AdamwTensor * len(chunk))()
        for j, t in enumerate(chunk):
            _chk(t, "tensor")
            arr[j] = _lib.AdamwTensor(None, t.data_ptr(), None, None, t.numel(), 0.0, 0.0)
        _lib.check(lib.tt_scale_tensors(arr, len(chunk), _p(scale), _stream()), "tt_scale_tensors")


def ema_update_(teacher, student, momentum: float):
    """teacher <- teacher * (1 - m) + student * m  (time_tuning.py:113-115)."""
    _bump_param_epoch()
    lib = _lib.load()
    _chk(teacher, "teacher"); _chk(student, "student")
    assert teacher.numel() == student.numel()
    _lib.check(lib.tt_ema_update(_p(teacher), _p(student), teacher.numel(), float(momentum), _stream()), "tt_ema_update")
    return teacher


def add_(dst, src):
    lib = _lib.load()
    _chk(dst, "dst"); _chk(src, "src")
    _lib.check(lib.tt_add_inplace(_p(dst), _p(src), dst.numel(), _stream()), "tt_add_inplace")
    return dst


def count_mismatch(a, b) -> int:
    """Number of positions where two fp32 buffers differ bitwise; synchronises (a one-off check, not a step op)."""
    lib = _lib.load()
    _chk(a, "a"); _chk(b, "b")
    assert a.numel() == b.numel()
    out = torch.empty((1,), dtype=torch.int64, device=a.device)
    _lib.check(lib.tt_count_mismatch(_p(a), _p(b), a.numel(), _p(out), _stream()), "tt_count_mismatch")
    return int(out.item())


# ---- bf16-plane operands (include/timetuning_hip.h: "bf16-plane operands") ------------------------------------------------
bf16 = torch.bfloat16


def split_planes(x, planes: int, out=None):
    """fp32 tensor -> [planes, *x.shape] bf16 with x = sum of the planes (exactly, for planes = 3)."""
    lib = _lib.load()
    _chk(x, "x")
    n = x.numel()
    if n % 8:
        raise ValueError("split_planes: the element count must be a multiple of 8")
    y = out if out is not None else torch.empty((planes, *x.shape), dtype=bf16, device=x.device)
    _lib.check(lib.tt_split_planes(_p(x), _p(y), n, int(planes), n, _stream()), "tt_split_planes")
    return y


def layernorm_fwd_planes(x, gamma, beta, planes: int, eps=1e-6, save_stats=False, drop_first_token=False):
    """LayerNorm whose result is written as bf16 planes [planes, rows, D] (see ``layernorm_fwd`` for the arguments)."""
    lib = _lib.load()
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta")
    D = x.shape[-1]
    rows = x.numel() // D
    skip = 0
    if drop_first_token:
        skip = x.shape[-2]
        rows = rows // skip * (skip - 1)
    y = torch.empty((planes, rows, D), dtype=bf16, device=x.device)
    mean = torch.empty((rows,), dtype=f32, device=x.device) if save_stats else None
    rstd = torch.empty((rows,), dtype=f32, device=x.device) if save_stats else None
    _lib.check(lib.tt_layernorm_fwd_planes(_p(x), _p(gamma), _p(beta), _p(y), rows * D, int(planes), _p(mean), _p(rstd), rows, D, float(eps),
                                           skip, _stream()), "tt_layernorm_fwd_planes")
    return (y, mean, rstd) if save_stats else y


def linear_fwd_planes(xp, wp, bias=None, residual=None, act: int = 0, out_f32: bool = True, out_planes: int = 0, save_pre: bool = False,
                      out=None):
    """y = act(x @ w.T + bias) (+ residual) on bf16-plane operands xp [P, M, K], wp [P, N, K].  Returns a dict with the
    requested outputs: ``y`` (fp32 [M,N]), ``planes`` (bf16 [out_planes, M, N]), ``pre`` (fp32 pre-activation)."""
    lib = _lib.load()
    _chk(xp, "xp", bf16); _chk(wp, "wp", bf16)
    P, M, K = xp.shape
    N = wp.shape[1]
    assert wp.shape[0] == P and wp.shape[2] == K, (xp.shape, wp.shape)
    if bias is not None: _chk(bias, "bias")
    if residual is not None: _chk(residual, "residual")
    y = (out if out is not None else torch.empty((M, N), dtype=f32, device=xp.device)) if out_f32 else None
    yp = torch.empty((out_planes, M, N), dtype=bf16, device=xp.device) if out_planes else None
    pre = torch.empty((M, N), dtype=f32, device=xp.device) if save_pre else None
    # (profiled runs book the launch under the kernel that really runs it)
    p8 = PROFILE is not None and lib.tt_linear_fwd_planes_route(P, M, N, K, int(act), int(bias is not None), int(residual is not None),
                                                                int(y is not None), int(out_planes), int(pre is not None)) == 8
    e0 = _prof_begin()
    kws = ksplit_workspace(xp.device)
    _lib.check(lib.tt_linear_fwd_planes(_p(xp), M * K, _p(wp), N * K, P, _p(bias), _p(residual), _p(y), _p(pre), _p(yp), M * N, int(out_planes),
                                        M, N, K, int(act), _p(kws), kws.numel(), _stream()), "tt_linear_fwd_planes")
    _prof_end(e0, f"PLANES8_{P}" if p8 else f"PLANES{P}", M, N, K)
    return dict(y=y, planes=yp, pre=pre)


_RANGE_PENDING: dict = {}


def poll_pair_range(device=None) -> None:
    """``check_pair_range`` without a synchronisation, for the per-step path: the flag travels to a pinned host word asynchronously, and the
    call raises PairRangeError when a PREVIOUS poll's copy has landed with the flag set (detection lags by one step)."""
    f = range_flag(device)
    if f is None:
        return
    idx = f.device.index
    pend = _RANGE_PENDING.get(idx)
    if pend is None:
        pend = _RANGE_PENDING[idx] = dict(host=torch.zeros(4, dtype=torch.int32).pin_memory(), event=None)
    if pend["event"] is not None and pend["event"].query():
        pend["event"] = None
        if int(pend["host"][0]) != 0:
            check_pair_range(device)   # (synchronises, resets, raises)
    if pend["event"] is None:
        pend["host"].copy_(f, non_blocking=True)
        pend["event"] = torch.cuda.Event()
        pend["event"].record()


# ---- fp16-pair operands (include/timetuning_hip.h: "fp16-PAIR operands") ------------------------------------------------------
f16 = torch.float16


def split_pairs(x, out=None):
    """fp32 tensor [..., C] (C % 32 == 0) -> fp16 pairs [..., 2 C]: groups of 32 elements as [hi x 32][lo x 32]."""
    lib = _lib.load()
    _chk(x, "x")
    if x.shape[-1] % 32:
        raise ValueError("split_pairs: the last dimension must be a multiple of 32")
    y = out if out is not None else torch.empty((*x.shape[:-1], 2 * x.shape[-1]), dtype=f16, device=x.device)
    _lib.check(lib.tt_split_pairs(_p(x), _p(y), x.numel(), _p(range_flag(x.device)), _stream()), "tt_split_pairs")
    return y


def join_pairs(xp):
    """fp16 pairs [..., 2 C] -> fp32 [..., C] (hi + lo 2^-11)."""
    lib = _lib.load()
    _chk(xp, "xp", f16)
    y = torch.empty((*xp.shape[:-1], xp.shape[-1] // 2), dtype=f32, device=xp.device)
    _lib.check(lib.tt_join_pairs(_p(xp), _p(y), y.numel(), _stream()), "tt_join_pairs")
    return y


def layernorm_fwd_pairs(x, gamma, beta, eps=1e-6, save_stats=False, drop_first_token=False):
    """LayerNorm whose result is written as fp16 pairs [rows, 2 D] (see ``layernorm_fwd`` for the arguments)."""
    lib = _lib.load()
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta")
    D = x.shape[-1]
    rows = x.numel() // D
    skip = 0
    if drop_first_token:
        skip = x.shape[-2]
        rows = rows // skip * (skip - 1)
    y = torch.empty((rows, 2 * D), dtype=f16, device=x.device)
    mean = torch.empty((rows,), dtype=f32, device=x.device) if save_stats else None
    rstd = torch.empty((rows,), dtype=f32, device=x.device) if save_stats else None
    _lib.check(lib.tt_layernorm_fwd_pairs(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, float(eps), skip,
                                          _p(range_flag(x.device)), _stream()), "tt_layernorm_fwd_pairs")
    return (y, mean, rstd) if save_stats else y


def linear_fwd_pairs(xp, wp, bias=None, residual=None, act: int = 0, out_f32: bool = True, out_pairs: bool = False, save_pre: bool = False,
                     out=None):
    """y = act(x @ w.T + bias) (+ residual) on fp16-pair operands xp [M, 2 K], wp [N, 2 K].  Returns a dict with the requested
    outputs: ``y`` (fp32 [M,N]), ``pairs`` (fp16 [M, 2 N]), ``pre`` (fp32 pre-activation)."""
    lib = _lib.load()
    _chk(xp, "xp", f16); _chk(wp, "wp", f16)
    M, K2 = xp.shape
    N = wp.shape[0]
    K = K2 // 2
    assert wp.shape[1] == K2, (xp.shape, wp.shape)
    if bias is not None: _chk(bias, "bias")
    if residual is not None: _chk(residual, "residual")
    y = (out if out is not None else torch.empty((M, N), dtype=f32, device=xp.device)) if out_f32 else None
    yp = torch.empty((M, 2 * N), dtype=f16, device=xp.device) if out_pairs else None
    pre = torch.empty((M, N), dtype=f32, device=xp.device) if save_pre else None
    p8 = PROFILE is not None and lib.tt_linear_fwd_pairs_route(M, N, K, int(act), int(bias is not None), int(residual is not None),
                                                               int(y is not None), int(yp is not None), int(pre is not None)) == 8
    e0 = _prof_begin()
    kws = ksplit_workspace(xp.device)
    _lib.check(lib.tt_linear_fwd_pairs(_p(xp), _p(wp), _p(bias), _p(residual), _p(y), _p(pre), _p(yp), M, N, K, int(act), _p(kws), kws.numel(),
                                       _p(range_flag(xp.device)) if yp is not None else None, _stream()), "tt_linear_fwd_pairs")
    _prof_end(e0, "PAIRS8" if p8 else "PAIRS", M, N, K)
    return dict(y=y, pairs=yp, pre=pre)


def attention_pairs_ok(N: int, head_dim: int) -> bool:
    """Shapes ``attention_fwd_pairs`` takes (any N: K / V resident in LDS up to 256 tokens, the KV-tiled kernel beyond)."""
    return head_dim == 64


def attention_fwd_pairs(qkv_pairs, num_heads: int, out_pairs: bool = True, out_f32: bool = False, save_lse: bool = False):
    """qkv [F, N, 2 * 3D] fp16 pairs -> (out pairs [F, N, 2 D] or None, out fp32 [F, N, D] or None, lse [F, H, N] or None)."""
    lib = _lib.load()
    _chk(qkv_pairs, "qkv_pairs", f16)
    F, N, D6 = qkv_pairs.shape
    D = D6 // 6
    hd = D // num_heads
    op = torch.empty((F, N, 2 * D), dtype=f16, device=qkv_pairs.device) if out_pairs else None
    of = torch.empty((F, N, D), dtype=f32, device=qkv_pairs.device) if out_f32 else None
    lse = torch.empty((F, num_heads, N), dtype=f32, device=qkv_pairs.device) if save_lse else None
    _lib.check(lib.tt_attention_fwd_pairs(_p(qkv_pairs), _p(op), _p(of), _p(lse), F, N, num_heads, hd, float(hd ** -0.5), _stream()),
               "tt_attention_fwd_pairs")
    return op, of, lse


def split_pairs_dual(x, want_row: bool = False, want_colsum: bool = False, rpad: Optional[int] = None, colsum_out=None, want_t: bool = True,
                     scaled: bool = False, colsum_parts: bool = False, amax_in=None):
    """fp32 [R, C] -> (transposed pairs [C, 2 Rpad] or None, row-major pairs [R, 2 C] or None, column sums [C] or None) in ONE pass
    (tt_split_pairs_dual): what the backward of an nn.Linear needs of its dy.  ``scaled`` (a GRADIENT, whose whole magnitude may sit
    below fp16's normal range): the pairs hold x * S with S the power of two that brings max |x| into [2^13, 2^14); a fourth value is
    returned, the DEVICE scalar S, which the products take as ``dy_scale`` and divide by (no host round trip).
    ``colsum_parts`` (tt_split_pairs_dual_parts): the column sums come back UNFOLDED, as the partials of 64-row blocks [ceil(Rpad / 64), C],
    for ``linear_bwd_weight_pairs_tn(..., colsum_parts=, db_out=)`` to fold in the launch that folds the weight gradient."""
    lib = _lib.load()
    _chk(x, "x")
    R, Cc = x.shape
    rpad = (R + 31) // 32 * 32 if rpad is None else rpad
    assert want_t or want_row
    if colsum_parts:
        assert want_colsum and colsum_out is None
        t = torch.empty((Cc, 2 * rpad), dtype=f16, device=x.device) if want_t else None
        row = torch.empty((R, 2 * Cc), dtype=f16, device=x.device) if want_row else None
        parts = torch.empty(((rpad + 63) // 64, Cc), dtype=f32, device=x.device)
        own_max = scaled and amax_in is None
        nb = lib.tt_split_pairs_dual_workspace_bytes(R, Cc, rpad) if own_max else 0
        ws = _ws(nb, x.device) if own_max else None
        scale = torch.empty((1,), dtype=f32, device=x.device) if scaled else None
        _lib.check(lib.tt_split_pairs_dual_parts(_p(x), _p(t), _p(row), _p(parts), _p(scale), _p(amax_in) if scaled else None, R, Cc, rpad, _p(ws), nb,
                                                 _p(range_flag(x.device)), _stream()), "tt_split_pairs_dual_parts")
        return (t, row, parts, scale) if scaled else (t, row, parts)
    t = torch.empty((Cc, 2 * rpad), dtype=f16, device=x.device) if want_t else None
    row = torch.empty((R, 2 * Cc), dtype=f16, device=x.device) if want_row else None
    sums = (_chk(colsum_out, "colsum_out") if colsum_out is not None else torch.empty((Cc,), dtype=f32, device=x.device)) if want_colsum else None
    need_ws = want_colsum or scaled
    nb = lib.tt_split_pairs_dual_workspace_bytes(R, Cc, rpad) if need_ws else 0
    ws = _ws(nb, x.device) if need_ws else None
    scale = torch.empty((1,), dtype=f32, device=x.device) if scaled else None
    _lib.check(lib.tt_split_pairs_dual(_p(x), _p(t), _p(row), _p(sums), _p(scale), R, Cc, rpad, _p(ws), nb, _p(range_flag(x.device)), _stream()),
               "tt_split_pairs_dual")
    return (t, row, sums, scale) if scaled else (t, row, sums)


def split_pairs_dual_multi(items) -> None:
    """``items``: (x fp32 [R, C], row pairs out [R, 2 C] or None, transposed pairs out [C, 2 Rpad] or None) - every matrix converted in ONE
    launch per 32 of them (tt_split_pairs_dual_multi): the pair operands of all the weights an optimizer / EMA update rewrote."""
    import ctypes as C
    lib = _lib.load()
    n = len(items)
    if n == 0:
        return
    src, dt, dr = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    Rs, Cs, Rp = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
    for i, (x, row, t) in enumerate(items):
        _chk(x, "x")
        R, Cc = x.shape
        rpad = (R + 31) // 32 * 32
        if row is not None:
            _chk(row, "row", f16); assert row.shape == (R, 2 * Cc), (row.shape, x.shape)
        if t is not None:
            _chk(t, "t", f16); assert t.shape == (Cc, 2 * rpad), (t.shape, x.shape)
        src[i], dr[i], dt[i] = _p(x), _p(row), _p(t)
        Rs[i], Cs[i], Rp[i] = R, Cc, rpad
    _lib.check(lib.tt_split_pairs_dual_multi(src, dt, dr, Rs, Cs, Rp, n, _p(range_flag()), _stream()), "tt_split_pairs_dual_multi")


def transpose_pairs(xp, rpad: Optional[int] = None):
    """pairs [R, 2 C] -> transposed pairs [C, 2 Rpad] (zero beyond R)."""
    lib = _lib.load()
    _chk(xp, "xp", f16)
    R, C2 = xp.shape
    rpad = (R + 31) // 32 * 32 if rpad is None else rpad
    t = torch.empty((C2 // 2, 2 * rpad), dtype=f16, device=xp.device)
    _lib.check(lib.tt_transpose_pairs(_p(xp), _p(t), R, C2 // 2, rpad, _stream()), "tt_transpose_pairs")
    return t


def bwd_pairs_ok(M: int, N: int, K: int) -> bool:
    """Shapes the pair backward products take (dy [M,N], w [N,K])."""
    return N % 64 == 0 and K % 64 == 0


def bwd_weight_pairs_tn_ok(M: int, N: int, K: int) -> bool:
    """Shapes the transpose-free weight-gradient kernel takes (dy [M,N], x [M,K] in row pairs)."""
    return bool(TN_WGRAD) and bool(_lib.load().tt_linear_bwd_weight_pairs_tn_ok(N, K, M))


def linear_bwd_weight_pairs_tn(dy_pairs, x_pairs, dw_out=None, dy_scale=None, colsum_parts=None, db_out=None):
    """dw [N,K] = dy^T @ x from ROW pairs dy [M, 2 N], x [M, 2 K] (gemm_pairs_tn.hip: no transposed copies); ``dy_scale``: the device
    scalar a scaled split of dy returned.  ``colsum_parts`` (``split_pairs_dual(..., colsum_parts=True)``): the bias gradient is folded
    from them in the same launch as dw's split partials and returned too: (dw, db)."""
    lib = _lib.load()
    _chk(dy_pairs, "dy_pairs", f16); _chk(x_pairs, "x_pairs", f16)
    M, N, K = dy_pairs.shape[0], dy_pairs.shape[1] // 2, x_pairs.shape[1] // 2
    assert x_pairs.shape[0] == M, (dy_pairs.shape, x_pairs.shape)
    dw = _chk(dw_out, "dw_out") if dw_out is not None else torch.empty((N, K), dtype=f32, device=dy_pairs.device)
    nb = lib.tt_linear_bwd_weight_pairs_tn_workspace_bytes(N, K, M)
    ws = _ws(nb, dy_pairs.device)
    e0 = _prof_begin()
    if colsum_parts is not None:
        _chk(colsum_parts, "colsum_parts")
        assert colsum_parts.shape[1] == N, (colsum_parts.shape, N)
        db = _chk(db_out, "db_out") if db_out is not None else torch.empty((N,), dtype=f32, device=dy_pairs.device)
        _lib.check(lib.tt_linear_bwd_weight_pairs_tn_bias(_p(dy_pairs), _p(x_pairs), _p(dw), _p(dy_scale), N, K, M, _p(ws), nb, _p(colsum_parts),
                                                          colsum_parts.shape[0], _p(db), _stream()), "tt_linear_bwd_weight_pairs_tn_bias")
        _prof_end(e0, "PAIRS_TN", N, K, M)
        return dw, db
    _lib.check(lib.tt_linear_bwd_weight_pairs_tn(_p(dy_pairs), _p(x_pairs), _p(dw), _p(dy_scale), N, K, M, _p(ws), nb, _stream()), "tt_linear_bwd_weight_pairs_tn")
    _prof_end(e0, "PAIRS_TN", N, K, M)
    return dw


# Weight gradients on a side stream (round 6): ``wgrad_fork(stream)`` at the top of a backward, ``wgrad_join()`` before anything reads a
# gradient (the optimizer, a gradient bucket's all-reduce).  Between them ``linear_bwd_pairs`` launches its transpose-free weight-gradient
# product on ``stream`` behind the dy's split and carries on with the data gradient on the current stream.
_WGRAD = {"side": None, "keep": []}


def wgrad_fork(stream) -> None:
    _WGRAD["side"] = stream


def wgrad_side():
    """The armed weight-gradient side stream, or None."""
    return _WGRAD["side"] if PROFILE is None else None


def wgrad_join(final: bool = True) -> None:
    side = _WGRAD["side"]
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
        _WGRAD["keep"].clear()
        if final:
            _WGRAD["side"] = None


def wgrad_call(fn, keep=()):
    """``fn()`` on the weight-gradient side stream when one is armed (``wgrad_fork``), behind everything queued on the current stream so far;
    ``keep``: the tensors ``fn`` reads that the caller is about to drop (held until the join)."""
    side = _WGRAD["side"] if PROFILE is None else None
    if side is None:
        return fn()
    side.wait_stream(torch.cuda.current_stream())
    _WGRAD["keep"].append(keep)
    with torch.cuda.stream(side):
        return fn()


def linear_bwd_data_pairs_is_persistent(M: int, N: int, K: int) -> bool:
    """Would the data gradient dx [M, K] = dy [M, N] @ w of an nn.Linear(K -> N) run on the persistent pair kernel (plain fp32 epilogue)?"""
    return _lib.load().tt_linear_fwd_pairs_route(M, K, N, 0, 0, 0, 1, 0, 0) == 8


def linear_bwd_pairs(dy, wT_pairs, xT_pairs=None, gelu_pre=None, need_bias: bool = True, need_dx: bool = True, dw_out=None, db_out=None, x_pairs=None,
                     dy_amax=None, dx_amax_out=None):
    """(dx, dw, db) of an nn.Linear on pair operands: dy fp32 [M,N] is split here in ONE pass (row pairs for both gradient products, column
    sums = the bias gradient); wT_pairs [K, 2 N] = the weight transposed in pairs.  The layer's input comes as ``x_pairs`` [M, 2 K] (row
    pairs as the forward kept them: the transpose-free weight-gradient kernel) or, for the shapes that kernel does not take, as
    ``xT_pairs`` [K, 2 Mpad] (``transpose_pairs``; dy's transposed pairs are then made in the same pass).
    ``dy_amax``: max |dy| as the kernel that produced dy published it (``AmaxPool``; the scaled split then skips its max pass);
    ``dx_amax_out``: a zeroed slot the data-gradient kernel raises to max |dx| (dx = the next Linear's dy: the gelu' route)."""
    lib = _lib.load()
    _chk(dy, "dy"); _chk(wT_pairs, "wT_pairs", f16)
    M, N = dy.shape
    K = wT_pairs.shape[0]
    if x_pairs is not None and bwd_weight_pairs_tn_ok(M, N, K):
        _chk(x_pairs, "x_pairs", f16)
        assert wT_pairs.shape == (K, 2 * N) and x_pairs.shape == (M, 2 * K), (dy.shape, wT_pairs.shape, x_pairs.shape)
        # the bias gradient's fold rides on the weight gradient's (one launch less per Linear): the column partials stay unfolded here
        r4 = split_pairs_dual(dy, want_row=True, want_colsum=need_bias, want_t=False, scaled=GRAD_SCALE, colsum_parts=need_bias,
                              amax_in=dy_amax if (need_bias and GRAD_SCALE) else None)
        dy_row, parts, dy_scale = r4[1], r4[2], (r4[3] if GRAD_SCALE else None)
        # the weight gradient is a LEAF of the backward (nothing downstream reads it before the optimizer): with a side stream set
        # (``wgrad_fork``: engine.TWO_STREAMS) it runs there, beside the data-gradient chain that continues on this stream
        # (kept alive until the join: freed earlier, their blocks could be reused on THIS stream under the side stream's reads)
        if need_bias:
            dw, db = wgrad_call(lambda: linear_bwd_weight_pairs_tn(dy_row, x_pairs, dw_out, dy_scale, colsum_parts=parts, db_out=db_out),
                                keep=(dy_row, parts, dy_scale, x_pairs))
        else:
            dw, db = wgrad_call(lambda: linear_bwd_weight_pairs_tn(dy_row, x_pairs, dw_out, dy_scale), keep=(dy_row, dy_scale, x_pairs)), None
    else:
        if xT_pairs is None:
            xT_pairs = transpose_pairs(x_pairs)
        _chk(xT_pairs, "xT_pairs", f16)
        Mpad = xT_pairs.shape[1] // 2
        assert wT_pairs.shape == (K, 2 * N) and xT_pairs.shape[0] == K and Mpad >= M, (dy.shape, wT_pairs.shape, xT_pairs.shape)
        r4 = split_pairs_dual(dy, want_row=need_dx, want_colsum=need_bias, rpad=Mpad, colsum_out=db_out, scaled=GRAD_SCALE)
        dyT, dy_row, db, dy_scale = r4[0], r4[1], r4[2], (r4[3] if GRAD_SCALE else None)
        dw = _chk(dw_out, "dw_out") if dw_out is not None else torch.empty((N, K), dtype=f32, device=dy.device)
        nb = lib.tt_linear_bwd_weight_pairs_workspace_bytes(N, K, Mpad)
        ws = _ws(nb, dy.device)
        e0 = _prof_begin()
        _lib.check(lib.tt_linear_bwd_weight_pairs(_p(dyT), _p(xT_pairs), _p(dw), _p(dy_scale), N, K, Mpad, _p(ws), nb, _stream()), "tt_linear_bwd_weight_pairs")
        _prof_end(e0, "PAIRS", N, K, M)
    dx = None
    if need_dx:
        if gelu_pre is not None: _chk(gelu_pre, "gelu_pre")
        dx = torch.empty((M, K), dtype=f32, device=dy.device)
        # (label: the persistent kernel takes a data-gradient product under the shape rules of its fp32 (+ operand) epilogues)
        p8 = PROFILE is not None and lib.tt_linear_fwd_pairs_route(M, K, N, 0, 0, int(gelu_pre is not None), 1, 0, 0) == 8
        e0 = _prof_begin()
        kws = ksplit_workspace(dx.device)
        _lib.check(lib.tt_linear_bwd_data_pairs(_p(dy_row), _p(wT_pairs), _p(gelu_pre), _p(dx), _p(dy_scale), M, N, K, _p(kws), kws.numel(),
                                                _p(dx_amax_out), _stream()), "tt_linear_bwd_data_pairs")
        _prof_end(e0, "PAIRS8" if p8 else "PAIRS", M, K, N)
    return dx, dw, db


def attention_fwd_bf16(qkv, num_heads: int):
    """qkv [F,N,3*D] bf16 -> out [F,N,D] bf16 (N <= 256, head_dim 64)."""
    lib = _lib.load()
    _chk(qkv, "qkv", bf16)
    F, N, D3 = qkv.shape
    D = D3 // 3
    hd = D // num_heads
    out = torch.empty((F, N, D), dtype=bf16, device=qkv.device)
    _lib.check(lib.tt_attention_fwd_bf16(_p(qkv), _p(out), F, N, num_heads, hd, float(hd ** -0.5), _stream()), "tt_attention_fwd_bf16")
    return out


def transpose_planes(x, rpad: Optional[int] = None, colsum: bool = False):
    """fp32 [R, C] -> bf16 [1, C, Rpad] (transposed, zero-padded along R to a multiple of 64 by default).  ``colsum``: the same pass also
    returns x.sum(0) in fp32 (tt_transpose_planes_colsum) -> (planes, sums)."""
    lib = _lib.load()
    _chk(x, "x")
    R, Cc = x.shape
    rpad = (R + 63) // 64 * 64 if rpad is None else rpad
    y = torch.empty((1, Cc, rpad), dtype=bf16, device=x.device)
    if colsum:
        sums = torch.empty((Cc,), dtype=f32, device=x.device)
        nb = lib.tt_transpose_planes_colsum_workspace_bytes(R, Cc, rpad)
        ws = _ws(nb, x.device)
        _lib.check(lib.tt_transpose_planes_colsum(_p(x), _p(y), R, Cc, rpad, _p(sums), _p(ws), nb, _stream()), "tt_transpose_planes_colsum")
        return y, sums
    _lib.check(lib.tt_transpose_planes(_p(x), _p(y), R, Cc, rpad, _stream()), "tt_transpose_planes")
    return y


def bwd_planes_ok(M: int, N: int, K: int) -> bool:
    """Shapes the bf16 backward products take (dy [M,N], w [N,K]): whole 64-wide tiles along N and K."""
    return N % 64 == 0 and K % 64 == 0


def linear_bwd_data_planes(dy, w, gelu_pre=None):
    """dx = dy @ w (* gelu'(gelu_pre)) on bf16 operands: dy [M,N] and w [N,K] fp32 are converted here (dy rounded, w
    transposed), fp32 accumulate and output."""
    lib = _lib.load()
    _chk(dy, "dy"); _chk(w, "w")
    M, N = dy.shape
    K = w.shape[1]
    dyp = split_planes(dy, 1)
    wT = transpose_planes(w, N)                      # [1, K, N]
    dx = torch.empty((M, K), dtype=f32, device=dy.device)
    if gelu_pre is not None: _chk(gelu_pre, "gelu_pre")
    e0 = _prof_begin()
    kws = ksplit_workspace(dx.device)
    _lib.check(lib.tt_linear_bwd_data_planes(_p(dyp), M * N, _p(wT), K * N, 1, _p(gelu_pre), _p(dx), M, N, K, _p(kws), kws.numel(), _stream()),
               "tt_linear_bwd_data_planes")
    _prof_end(e0, "PLANES1", M, K, N)
    return dx


def linear_bwd_weight_planes(dy, x, need_bias=True):
    """dw = dy.T @ x on bf16 operands (both transposed + rounded here), db = dy.sum(0) in fp32."""
    lib = _lib.load()
    _chk(dy, "dy"); _chk(x, "x")
    M, N = dy.shape
    K = x.shape[1]
    # [1, N, Mpad], [1, K, Mpad]; the bias gradient dy.sum(0) comes out of dy's transposing pass
    dyT, db = transpose_planes(dy, colsum=True) if need_bias else (transpose_planes(dy), None)
    xT = transpose_planes(x)
    Mpad = dyT.shape[2]
    dw = torch.empty((N, K), dtype=f32, device=dy.device)
    nb = lib.tt_linear_bwd_weight_planes_workspace_bytes(N, K, Mpad)
    ws = _ws(nb, dy.device)
    e0 = _prof_begin()
    _lib.check(lib.tt_linear_bwd_weight_planes(_p(dyT), N * Mpad, _p(xT), K * Mpad, 1, _p(dw), N, K, Mpad, _p(ws), nb, _stream()),
               "tt_linear_bwd_weight_planes")
    _prof_end(e0, "PLANES1", N, K, Mpad)
    return dw, db


# ---- coarse entry points: one C call per reference function ------------------------------------------------

def fine_grained() -> bool:
    """True while a per-launch GEMM profile is being collected (bench.py's roofline leg): the launch sequences then go through
    the op-level entry points one ctypes call at a time - the same kernels in the same order as the coarse entry points
    enqueue - so that every GEMM launch can be bracketed by events."""
    return PROFILE is not None


def vit_forward(params, n_blocks: int, tokens, img=None, frame_map=None, normed_out=None, drop_cls: bool = False, last_qkv: bool = False,
                last_probs: bool = False):
    """``tt_vit_forward``: [prepare_tokens of ``img``] + ``n_blocks`` blocks in place on ``tokens`` [F,N,D] [+ final norm].
    ``params`` = (VitParams struct, keep-alive list) as ``engine.vit_params`` builds it.  Returns (normed or None, qkv of the
    last block or None, its attention probabilities or None)."""
    lib = _lib.load()
    vp, _keep = params
    _chk(tokens, "tokens")
    F, N, D = tokens.shape
    vp.n_blocks = int(n_blocks)
    planes_given = vp.planes   # (the caller's struct is restored below - a cached (vp, keep) must not stay in f32 for later launches)
    if vp.planes == 2 and F * N < PAIRS_MIN_ROWS:
        vp.planes = 0   # a small launch sequence keeps the exact-f32 kernels (PAIRS_MIN_ROWS); the fp32 weights are in the table anyway
    dev = tokens.device
    rf = range_flag(dev) if vp.planes == 2 else None
    vp.range_flag = _p(rf)
    vp.precision = precision_code()
    if img is not None:
        _chk(img, "img")
        C_, H, W = img.shape[1], img.shape[2], img.shape[3]
    else:
        # the token grid only matters through N: hand the entry point an H x W that yields it
        C_, H, W = 3, vp.patch, (N - 1) * vp.patch
    normed = None
    if normed_out is not None:
        normed = normed_out if isinstance(normed_out, torch.Tensor) else torch.empty((F * (N - 1), D) if drop_cls else (F, N, D), dtype=f32, device=dev)
        _chk(normed, "normed_out")
    qkv = torch.empty((F, N, 3 * D), dtype=f32, device=dev) if last_qkv else None
    probs = torch.empty((F, vp.heads, N, N), dtype=f32, device=dev) if last_probs else None
    nb = lib.tt_vit_forward_workspace_bytes(F, N, D, vp.hidden, vp.planes) if n_blocks else 0
    ws = _ws(nb, dev)
    try:
        _lib.check(lib.tt_vit_forward(vp, _p(img), _p(frame_map), F, C_, H, W, _p(tokens), _p(normed), int(bool(drop_cls)), _p(qkv), _p(probs),
                                      _p(ws), nb, _stream()), "tt_vit_forward")
    finally:
        vp.planes = planes_given
    return normed, qkv, probs


def mlp_head_forward(x, layers: Sequence[tuple]):
    """``tt_mlp_head_forward``: layers = [(weight [out,in], bias or None), ...], GELU between them."""
    lib = _lib.load()
    _chk(x, "x")
    M = x.shape[0]
    arr = (_lib.LinearParams * len(layers))()
    for i, (w, b) in enumerate(layers):
        _chk(w, "weight")
        if b is not None: _chk(b, "bias")
        arr[i] = _lib.LinearParams(w.data_ptr(), b.data_ptr() if b is not None else None, w.shape[0], w.shape[1])
    out = torch.empty((M, layers[-1][0].shape[0]), dtype=f32, device=x.device)
    nb = lib.tt_mlp_head_forward_workspace_bytes(M, arr, len(layers))
    ws = _ws(nb, x.device)
    _lib.check(lib.tt_mlp_head_forward(_p(x), M, arr, len(layers), _p(out), precision_code(), _p(ws), nb, _stream()), "tt_mlp_head_forward")
    return out


def scores_sinkhorn(z, prototypes, queue=None, iters: int = 10, eps: float = 0.05, rows_out: Optional[int] = None):
    """``tt_scores_sinkhorn``: (q [rows_out, K], scores [B + queue rows, K]) of TimeT.get_scores on one rank."""
    lib = _lib.load()
    _chk(z, "z"); _chk(prototypes, "prototypes")
    B, dim = z.shape
    K = prototypes.shape[0]
    Qr = 0
    if queue is not None:
        _chk(queue, "queue")
        Qr = queue.shape[0]
    rows_out = B if rows_out is None else rows_out
    scores = torch.empty((B + Qr, K), dtype=f32, device=z.device)
    q = torch.empty((rows_out, K), dtype=f32, device=z.device)
    nb = lib.tt_scores_sinkhorn_workspace_bytes(B, Qr, K, dim)
    ws = _ws(nb, z.device)
    _lib.check(lib.tt_scores_sinkhorn(_p(z), B, _p(queue), Qr, _p(prototypes), K, dim, _p(scores), _p(q), rows_out, float(eps), int(iters),
                                      precision_code(), _p(ws), nb, _stream()), "tt_scores_sinkhorn")
    return q, scores


def adamw_ema_step_(entries: Sequence[tuple], step: int, beta1=0.9, beta2=0.999, eps=1e-8, prototypes=None, teacher_flat=None,
                    student_flat=None, teacher_prototypes=None, momentum: float = 0.0):
    """``tt_adamw_ema_step``: AdamW over ``entries`` (as ``adamw_step_``), prototypes renormalised, then the EMA teacher update of
    the flat parameter buffers and the teacher prototypes (each part skipped when its tensors are None)."""
    _bump_param_epoch()
    lib = _lib.load()
    arr = (_lib.AdamwTensor * max(len(entries), 1))()
    for j, (p, g, m, v, lr, wd) in enumerate(entries):
        for t, nm in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            _chk(t, nm)
        arr[j] = _lib.AdamwTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(wd))
    K = dim = 0
    if prototypes is not None:
        _chk(prototypes, "prototypes")
        K, dim = prototypes.shape
    n_flat = 0
    if teacher_flat is not None:
        _chk(teacher_flat, "teacher_flat"); _chk(student_flat, "student_flat")
        assert teacher_flat.numel() == student_flat.numel()
        n_flat = teacher_flat.numel()
    if teacher_prototypes is not None:
        _chk(teacher_prototypes, "teacher_prototypes")
    _lib.check(lib.tt_adamw_ema_step(arr, len(entries), int(step), float(beta1), float(beta2), float(eps), _p(prototypes), K, dim,
                                     _p(teacher_flat) if n_flat else None, _p(student_flat) if n_flat else None, n_flat, _p(teacher_prototypes),
                                     float(momentum), _stream()), "tt_adamw_ema_step")
