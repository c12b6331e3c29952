"""Launch sequences of the TimeTuning hot path on top of the C ABI (hip_ops).

Nothing here computes: each function strings HIP kernels together on PyTorch's current stream and
keeps the device buffers alive.  Three sequences:

``vit_tokens``       patch-embed + 12 blocks (blocks flagged trainable keep their activations)
``extractor``        + final LayerNorm with the cls row dropped (+ projection head)
``TimeT._run_step``  (time_tuning.py of this package) the whole objective of ``TimeT.get_loss`` (time_tuning.py:224-302) with ONE
                     student pass per frame instead of the reference's four, the head only on the
                     source/target frames, one Sinkhorn solve instead of two (or four with a teacher),
                     batched label propagation without host round trips, fused CE forward/backward and
                     the backward of blocks.10/11 + head + prototypes on the target frames only.
                     SURVEY.md section 3.2/3.3 documents why these are result-preserving.

Frames are processed TIME-MAJOR ([fs, bs] instead of the input's [bs, fs]): the patch-embed kernel gathers
through a frame map (no copy), and the target frames then form one contiguous row range for backward.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import hip_ops as ops

f32 = torch.float32


# ------------------------------------------------------------------------------------------------
# ViT blocks
# ------------------------------------------------------------------------------------------------

def _param_tag(w: torch.Tensor, kind) -> tuple:
    """What a cached operand derived from ``w`` is valid for: the storage, torch's version counter (in-place edits through ATen) and the
    epoch of raw-pointer parameter updates (``hip_ops.PARAM_EPOCH``: AdamW / EMA / prototype renormalisation write through data_ptr and do
    not advance the version counter)."""
    # a tensor marked static (``_tt_static``: the student's frozen parameters, FeatureExtractor.freeze_backbone) and still frozen is never
    # rewritten through raw pointers: its operands are made once; staticness is re-derived at use time (a tensor unfrozen later is trainable)
    static = getattr(w, "_tt_static", False) and not w.requires_grad
    return (w.data_ptr(), w._version, 0 if static else ops.PARAM_EPOCH, kind)


def weight_planes(w: torch.nn.Parameter, planes: int) -> torch.Tensor:
    """[planes, N, K] bf16 planes (planes = 2: [N, 2 K] fp16 pairs, the "f16x3" mode) of a Linear weight, cached ON the parameter object
    (not in a table keyed by id(): a recycled id + recycled storage of a dead model's parameter would otherwise hit) against
    ``_param_tag``: a frozen tensor is split once, a trainable or EMA'd one once per optimizer / EMA update - not once per use (the two
    streams of a trainable block, the source and target rows of the head)."""
    hit = getattr(w, "_tt_planes", None)
    tag = _param_tag(w, planes)
    if hit is None or hit[0] != tag:
        hit = (tag, ops.split_pairs(w.detach()) if planes == 2 else ops.split_planes(w.detach(), planes))
        w._tt_planes = hit
    return hit[1]


def refresh_pair_operands(weights) -> int:
    """The fp16-pair operands of Linear weights in ONE launch: for every 2-D weight in ``weights`` whose cached operands are stale (the
    optimizer / EMA update rewrote it: ``_param_tag``) the row pairs (``weight_planes(w, 2)``) and, for a trainable one, the transposed pairs
    (``weight_pairs_t(w)``) are rewritten IN their existing buffers and the caches re-tagged - where the lazy per-use route costs two small
    launches per weight and step.  Only operands that exist already are refreshed (a weight's first use makes them lazily: what no kernel
    ever read in pairs is never converted).  Returns the number of weights converted."""
    items, fix = [], []
    for w in weights:
        if w.dim() != 2 or w.shape[1] % 32 != 0 or w.dtype != torch.float32:
            continue
        tag_row, tag_t = _param_tag(w, 2), _param_tag(w, "T")
        hit_row, hit_t = getattr(w, "_tt_planes", None), getattr(w, "_tt_pairs_t", None)
        need_row = hit_row is not None and hit_row[0][3] == 2 and hit_row[0] != tag_row
        need_t = hit_t is not None and hit_t[0] != tag_t
        if not (need_row or need_t):
            continue
        N, K = w.shape
        row = t = None
        if need_row:
            old = hit_row[1] if hit_row is not None else None
            row = old if (old is not None and old.dtype == torch.float16 and old.shape == (N, 2 * K)) else torch.empty((N, 2 * K), dtype=torch.float16, device=w.device)
        if need_t:
            npad = (N + 31) // 32 * 32
            old = hit_t[1] if hit_t is not None else None
            t = old if (old is not None and old.shape == (K, 2 * npad)) else torch.empty((K, 2 * npad), dtype=torch.float16, device=w.device)
        items.append((w.detach(), row, t))
        fix.append((w, tag_row if need_row else None, row, tag_t if need_t else None, t))
    ops.split_pairs_dual_multi(items)
    for w, tag_row, row, tag_t, t in fix:
        if tag_row is not None:
            w._tt_planes = (tag_row, row)
        if tag_t is not None:
            w._tt_pairs_t = (tag_t, t)
    return len(items)


def block_forward_planes(x: torch.Tensor, blk, num_heads: int, planes: int, aux: Optional[dict] = None) -> torch.Tensor:
    """A block that keeps nothing, on bf16-plane operands (hip_ops.set_gemm_precision "bf16" / "bf16x6"): every Linear reads
    planes its producer wrote (LayerNorm, the fc1 / attention epilogues) - no conversion on the GEMMs' path.  The residual
    stream x [F,N,D] stays fp32 and is updated in place.  planes = 1: bf16 qkv and the bf16 attention kernel; planes = 3: fp32
    qkv, the fp32 attention kernel, its output split for proj."""
    Fr, N, D = x.shape
    M = Fr * N
    x2d = x.view(M, D)
    at = blk.attn
    if planes == 2:   # fp16 pairs ("f16x3"): LayerNorm and the fc1 epilogue write pairs, the fp32 attention's output is split for proj
        h1 = ops.layernorm_fwd_pairs(x, blk.norm1.weight, blk.norm1.bias)
        if aux is None and ops.attention_pairs_ok(N, D // num_heads):
            # qkv never exists in fp32: the qkv epilogue writes pairs, the pair attention kernel reads them and writes proj's operand
            qkvp = ops.linear_fwd_pairs(h1, weight_planes(at.qkv.weight, 2), at.qkv.bias, out_f32=False, out_pairs=True)["pairs"]
            attp = ops.attention_fwd_pairs(qkvp.view(Fr, N, 6 * D), num_heads)[0].view(M, 2 * D)
        else:
            qkv = ops.linear_fwd_pairs(h1, weight_planes(at.qkv.weight, 2), at.qkv.bias)["y"]
            if aux is not None:
                aux["qkv"] = qkv.view(Fr, N, 3 * D)
            att32, _, _ = ops.attention_fwd(qkv.view(Fr, N, 3 * D), num_heads)
            attp = ops.split_pairs(att32.view(M, D))
        ops.linear_fwd_pairs(attp, weight_planes(at.proj.weight, 2), at.proj.bias, residual=x2d, out=x2d)
        h2 = ops.layernorm_fwd_pairs(x, blk.norm2.weight, blk.norm2.bias)
        a = ops.linear_fwd_pairs(h2, weight_planes(blk.mlp.fc1.weight, 2), blk.mlp.fc1.bias, act=1, out_f32=False, out_pairs=True)["pairs"]
        ops.linear_fwd_pairs(a, weight_planes(blk.mlp.fc2.weight, 2), blk.mlp.fc2.bias, residual=x2d, out=x2d)
        return x
    h1 = ops.layernorm_fwd_planes(x, blk.norm1.weight, blk.norm1.bias, planes)
    if planes == 1 and aux is None and N <= 256 and D // num_heads == 64:
        qkv = ops.linear_fwd_planes(h1, weight_planes(at.qkv.weight, 1), at.qkv.bias, out_f32=False, out_planes=1)["planes"]
        att = ops.attention_fwd_bf16(qkv.view(Fr, N, 3 * D), num_heads).view(1, M, D)
    else:
        qkv = ops.linear_fwd_planes(h1, weight_planes(at.qkv.weight, planes), at.qkv.bias)["y"]
        if aux is not None:
            aux["qkv"] = qkv.view(Fr, N, 3 * D)
        att32, _, _ = ops.attention_fwd(qkv.view(Fr, N, 3 * D), num_heads)
        att = ops.split_planes(att32.view(M, D), planes)
    ops.linear_fwd_planes(att, weight_planes(at.proj.weight, planes), at.proj.bias, residual=x2d, out=x2d)
    h2 = ops.layernorm_fwd_planes(x, blk.norm2.weight, blk.norm2.bias, planes)
    a = ops.linear_fwd_planes(h2, weight_planes(blk.mlp.fc1.weight, planes), blk.mlp.fc1.bias, act=1, out_f32=False, out_planes=planes)["planes"]
    ops.linear_fwd_planes(a, weight_planes(blk.mlp.fc2.weight, planes), blk.mlp.fc2.bias, residual=x2d, out=x2d)
    return x


def block_forward(x: torch.Tensor, blk, num_heads: int, save: Optional[dict] = None, aux: Optional[dict] = None) -> torch.Tensor:
    """One transformer block (dino_vision_transformer.py:147-153) on x [F,N,D].  With ``save`` the
    activations backward needs are kept there; otherwise the residual stream is updated in place.
    ``aux`` (a dict) receives the block's qkv activations [F,N,3D] - what the attention foreground mask reads."""
    Fr, N, D = x.shape
    M = Fr * N
    planes = ops.plane_count_for(M)
    if save is None and planes and D % 64 == 0 and blk.mlp.fc1.weight.shape[0] % 64 == 0:
        return block_forward_planes(x, blk, num_heads, planes, aux)
    x2d = x.view(M, D)
    if save is not None and planes == 2 and D % 64 == 0 and blk.mlp.fc1.weight.shape[0] % 64 == 0:
        return block_forward_pairs_kept(x, blk, num_heads, save, aux)
    if save is not None:
        h1, save["mean1"], save["rstd1"] = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, save_stats=True)
    else:
        h1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias)
    qkv = ops.linear_fwd(h1.view(M, D), blk.attn.qkv.weight, blk.attn.qkv.bias)
    if aux is not None:
        aux["qkv"] = qkv.view(Fr, N, 3 * D)
    att, lse, _ = ops.attention_fwd(qkv.view(Fr, N, 3 * D), num_heads, save_lse=save is not None)
    x_mid = ops.linear_fwd(att.view(M, D), blk.attn.proj.weight, blk.attn.proj.bias, residual=x2d,
                           out=None if save is not None else x2d)
    if save is not None:
        h2, save["mean2"], save["rstd2"] = ops.layernorm_fwd(x_mid, blk.norm2.weight, blk.norm2.bias, save_stats=True)
        a, pre = ops.linear_fwd(h2, blk.mlp.fc1.weight, blk.mlp.fc1.bias, act=1, save_pre=True)
    else:
        h2 = ops.layernorm_fwd(x_mid, blk.norm2.weight, blk.norm2.bias, out=h1.view(M, D))
        a = ops.linear_fwd(h2, blk.mlp.fc1.weight, blk.mlp.fc1.bias, act=1)
    x_out = ops.linear_fwd(a, blk.mlp.fc2.weight, blk.mlp.fc2.bias, residual=x_mid, out=None if save is not None else x_mid)
    if save is not None:
        save.update(x_in=x, h1=h1, qkv=qkv, att=att, lse=lse, x_mid=x_mid, h2=h2, pre=pre, a=a)
    return x_out.view(Fr, N, D)


def block_forward_pairs_kept(x: torch.Tensor, blk, num_heads: int, save: dict, aux: Optional[dict] = None) -> torch.Tensor:
    """A block that KEEPS its activations, in the "f16x3" mode: the Linears run on fp16-pair operands (three fp16 MFMAs per term) and what
    the backward needs of their inputs is kept AS pairs (h1, att, h2, gelu(fc1)) - the weight-gradient products read them transposed
    (``ops.transpose_pairs`` in ``block_backward``); the residual streams, qkv, the attention output / lse, the fc1 pre-activation and the
    LayerNorm statistics are kept in fp32 as in the f32 mode (LayerNorm / attention / GELU backward read them)."""
    Fr, N, D = x.shape
    M = Fr * N
    x2d = x.view(M, D)
    at, mlp = blk.attn, blk.mlp
    h1p, save["mean1"], save["rstd1"] = ops.layernorm_fwd_pairs(x, blk.norm1.weight, blk.norm1.bias, save_stats=True)
    if ops.attention_pairs_ok(N, D // num_heads):
        # qkv leaves the Linear in fp32 (the attention backward reads it) AND in pairs (the pair attention kernel's operand); the attention
        # writes its output in pairs (proj's operand) and in fp32 + lse (the backward's)
        o = ops.linear_fwd_pairs(h1p, weight_planes(at.qkv.weight, 2), at.qkv.bias, out_pairs=True)
        qkv = o["y"]
        attp, att, lse = ops.attention_fwd_pairs(o["pairs"].view(Fr, N, 6 * D), num_heads, out_f32=True, save_lse=True)
        attp = attp.view(M, 2 * D)
    else:
        qkv = ops.linear_fwd_pairs(h1p, weight_planes(at.qkv.weight, 2), at.qkv.bias)["y"]
        att, lse, _ = ops.attention_fwd(qkv.view(Fr, N, 3 * D), num_heads, save_lse=True)
        attp = ops.split_pairs(att.view(M, D))
    if aux is not None:
        aux["qkv"] = qkv.view(Fr, N, 3 * D)
    x_mid = ops.linear_fwd_pairs(attp, weight_planes(at.proj.weight, 2), at.proj.bias, residual=x2d)["y"]
    h2p, save["mean2"], save["rstd2"] = ops.layernorm_fwd_pairs(x_mid, blk.norm2.weight, blk.norm2.bias, save_stats=True)
    o = ops.linear_fwd_pairs(h2p, weight_planes(mlp.fc1.weight, 2), mlp.fc1.bias, act=1, out_f32=False, out_pairs=True, save_pre=True)
    x_out = ops.linear_fwd_pairs(o["pairs"], weight_planes(mlp.fc2.weight, 2), mlp.fc2.bias, residual=x_mid)["y"]
    save.update(x_in=x, h1p=h1p, qkv=qkv, att=att, attp=attp, lse=lse, x_mid=x_mid, h2p=h2p, pre=o["pre"], ap=o["pairs"], pairs=True)
    return x_out.view(Fr, N, D)


def weight_pairs_t(w: torch.Tensor) -> torch.Tensor:
    """[K, 2 N] fp16 pairs of w^T for an nn.Linear weight w [N, K]: the operand of the data-gradient product in the "f16x3" mode, cached
    per parameter update like ``weight_planes`` (and made in the same pass as the forward operand when that is not cached yet)."""
    hit = getattr(w, "_tt_pairs_t", None)
    tag = _param_tag(w, "T")
    if hit is None or hit[0] != tag:
        fwd = getattr(w, "_tt_planes", None)
        need_row = fwd is None or fwd[0] != _param_tag(w, 2)
        t, row, _ = ops.split_pairs_dual(w.detach(), want_row=need_row)
        if need_row:
            w._tt_planes = (_param_tag(w, 2), row)
        hit = (tag, t)
        w._tt_pairs_t = hit
    return hit[1]


def _bwd_both_pairs(dy: torch.Tensor, w: torch.Tensor, xp: torch.Tensor, gelu_pre: Optional[torch.Tensor] = None, need_dx: bool = True,
                    dw_out=None, db_out=None, dy_amax=None, dx_amax_out=None):
    """(dx, dw, db) of an nn.Linear on pair operands: xp = the layer's input as kept by the forward (row-major pairs [M, 2 K]).
    ``dy_amax`` / ``dx_amax_out``: the maxima the gradient producers publish for the pair splits (``ops.AmaxPool``)."""
    return ops.linear_bwd_pairs(dy, weight_pairs_t(w), gelu_pre=gelu_pre, need_dx=need_dx, dw_out=dw_out, db_out=db_out, x_pairs=xp,
                                dy_amax=dy_amax, dx_amax_out=dx_amax_out if need_dx else None)


def _bwd_weight(dy: torch.Tensor, x: torch.Tensor, need_bias: bool = True):
    """dw, db of an nn.Linear: the f32-MFMA kernels, or - in the "bf16" mode (BASELINE C4's path) - the bf16-plane products."""
    if ops.plane_count() == 1 and ops.bwd_planes_ok(dy.shape[0], dy.shape[1], x.shape[1]):
        return ops.linear_bwd_weight_planes(dy, x, need_bias)
    return ops.linear_bwd_weight(dy, x, need_bias=need_bias)


def _bwd_data(dy: torch.Tensor, w: torch.Tensor, gelu_pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    if ops.plane_count() == 1 and ops.bwd_planes_ok(dy.shape[0], dy.shape[1], w.shape[1]):
        return ops.linear_bwd_data_planes(dy, w.detach(), gelu_pre)
    return ops.linear_bwd_data(dy, w, gelu_pre=gelu_pre)


def _bwd_both(dy: torch.Tensor, w: torch.Tensor, x: torch.Tensor, gelu_pre: Optional[torch.Tensor] = None, dw_out=None, db_out=None):
    """(dx, dw, db) of an nn.Linear whose input gradient is needed too: one launch for both products in the f32 mode.  ``dw_out`` /
    ``db_out``: destinations in the exchange's flat buckets (the "bf16" mode's products allocate their own; the exchange copies those in)."""
    if ops.plane_count() == 1:
        dw, db = _bwd_weight(dy, x)
        return _bwd_data(dy, w, gelu_pre), dw, db
    return ops.linear_bwd(dy, w, x, gelu_pre=gelu_pre, dw_out=dw_out, db_out=db_out)


def block_backward(dx_out: torch.Tensor, blk, num_heads: int, sv: dict, f0: int, f1: int, grads: Dict[torch.nn.Parameter, torch.Tensor],
                   need_dx: bool = True, after_mlp=None, out=None, dx_out_amax=None, amax_pool=None):
    """Backward of one block restricted to frames [f0, f1) of the saved activations.  dx_out [(f1-f0)*N, D]
    is consumed (overwritten).  Writes parameter gradients into ``grads``.  ``after_mlp()`` is called once the MLP's
    gradients (two thirds of a block's parameters) exist - the data-parallel exchange sends them while the attention half runs.
    ``out(param)`` -> a caller-owned destination for that parameter's gradient or None (``GradExchange.out``: the kernels then write the
    exchange's flat buckets directly).  "f16x3" mode: ``amax_pool`` (``ops.AmaxPool``) hands out the slots in which the kernel that
    PRODUCES a dy leaves max |dy| for that dy's scaled pair split (no max pass per dy); ``dx_out_amax`` = that slot for ``dx_out``.
    Returns dx_in - with ``amax_pool``: (dx_in, its slot)."""
    take = amax_pool.take if amax_pool is not None else (lambda: None)
    Fr, N, D = sv["x_in"].shape
    r0, r1 = f0 * N, f1 * N
    out = out or (lambda p: None)
    o = lambda lin: dict(dw_out=out(lin.weight), db_out=out(lin.bias))
    oln = lambda ln: dict(dg_out=out(ln.weight), db_out=out(ln.bias))
    if sv.get("pairs"):   # the "f16x3" mode: the four Linears' backward products on pair operands
        a_fc1, a_proj, a_qkv, a_in = take(), take(), take(), take()
        d_pre, grads[blk.mlp.fc2.weight], grads[blk.mlp.fc2.bias] = _bwd_both_pairs(dx_out, blk.mlp.fc2.weight, sv["ap"][r0:r1], sv["pre"][r0:r1],
                                                                                     dy_amax=dx_out_amax, dx_amax_out=a_fc1, **o(blk.mlp.fc2))
        d_h2, grads[blk.mlp.fc1.weight], grads[blk.mlp.fc1.bias] = _bwd_both_pairs(d_pre, blk.mlp.fc1.weight, sv["h2p"][r0:r1], dy_amax=a_fc1,
                                                                                    **o(blk.mlp.fc1))
        dx_mid, grads[blk.norm2.weight], grads[blk.norm2.bias] = ops.layernorm_bwd(
            d_h2, sv["x_mid"][r0:r1], blk.norm2.weight, sv["mean2"][r0:r1], sv["rstd2"][r0:r1], dx_accum=dx_out, amax_out=a_proj, **oln(blk.norm2))
        if after_mlp is not None:
            after_mlp()
        att_pairs = ops.ATTN_BWD_PAIRS and D // num_heads == 64
        # max |d_att| for the pair attention backward's scale: left by the proj data gradient when that runs on the general kernel anyway
        # (the persistent kernel publishes a maximum from its gelu' epilogue only); otherwise the attention backward measures it itself
        a_att = take() if (att_pairs and amax_pool is not None and not ops.linear_bwd_data_pairs_is_persistent(r1 - r0, D, D)) else None
        d_att, grads[blk.attn.proj.weight], grads[blk.attn.proj.bias] = _bwd_both_pairs(dx_mid, blk.attn.proj.weight, sv["attp"][r0:r1],
                                                                                        dy_amax=a_proj, dx_amax_out=a_att, **o(blk.attn.proj))
        dqkv = ops.attention_bwd(sv["qkv"].view(Fr, N, 3 * D)[f0:f1], sv["att"][f0:f1], d_att.view(f1 - f0, N, D), sv["lse"][f0:f1], num_heads,
                                 amax_out=a_qkv, pair_products=att_pairs, dout_amax=a_att)
        d_h1, grads[blk.attn.qkv.weight], grads[blk.attn.qkv.bias] = _bwd_both_pairs(dqkv.view((f1 - f0) * N, 3 * D), blk.attn.qkv.weight,
                                                                                       sv["h1p"][r0:r1], dy_amax=a_qkv, **o(blk.attn.qkv))
        dx_in, grads[blk.norm1.weight], grads[blk.norm1.bias] = ops.layernorm_bwd(
            d_h1, sv["x_in"].view(Fr * N, D)[r0:r1], blk.norm1.weight, sv["mean1"][r0:r1], sv["rstd1"][r0:r1], dx_accum=dx_mid,
            amax_out=a_in if need_dx else None, **oln(blk.norm1))
        if amax_pool is not None:
            return (dx_in, a_in) if need_dx else (None, None)
        return dx_in if need_dx else None
    a, pre, h2 = sv["a"][r0:r1], sv["pre"][r0:r1], sv["h2"][r0:r1]
    # x_out = x_mid + fc2(gelu(fc1(ln2(x_mid))))
    d_pre, grads[blk.mlp.fc2.weight], grads[blk.mlp.fc2.bias] = _bwd_both(dx_out, blk.mlp.fc2.weight, a, pre, **o(blk.mlp.fc2))
    d_h2, grads[blk.mlp.fc1.weight], grads[blk.mlp.fc1.bias] = _bwd_both(d_pre, blk.mlp.fc1.weight, h2, **o(blk.mlp.fc1))
    dx_mid, grads[blk.norm2.weight], grads[blk.norm2.bias] = ops.layernorm_bwd(
        d_h2, sv["x_mid"][r0:r1], blk.norm2.weight, sv["mean2"][r0:r1], sv["rstd2"][r0:r1], dx_accum=dx_out, **oln(blk.norm2))
    if after_mlp is not None:
        after_mlp()
    # x_mid = x_in + proj(attention(qkv(ln1(x_in))))
    att = sv["att"].view(Fr * N, D)[r0:r1]
    d_att, grads[blk.attn.proj.weight], grads[blk.attn.proj.bias] = _bwd_both(dx_mid, blk.attn.proj.weight, att, **o(blk.attn.proj))
    qkv = sv["qkv"].view(Fr, N, 3 * D)[f0:f1]
    # (the "bf16" mode - BASELINE C4's path - runs the attention backward's products on bf16 MFMA like its dgrad / wgrad products)
    dqkv = ops.attention_bwd(qkv, sv["att"][f0:f1], d_att.view(f1 - f0, N, D), sv["lse"][f0:f1], num_heads,
                             bf16_products=ops.plane_count() == 1 and D // num_heads == 64)
    dqkv2 = dqkv.view((f1 - f0) * N, 3 * D)
    h1 = sv["h1"].view(Fr * N, D)[r0:r1]
    d_h1, grads[blk.attn.qkv.weight], grads[blk.attn.qkv.bias] = _bwd_both(dqkv2, blk.attn.qkv.weight, h1, **o(blk.attn.qkv))
    x_in = sv["x_in"].view(Fr * N, D)[r0:r1]
    dx_in, grads[blk.norm1.weight], grads[blk.norm1.bias] = ops.layernorm_bwd(
        d_h1, x_in, blk.norm1.weight, sv["mean1"][r0:r1], sv["rstd1"][r0:r1], dx_accum=dx_mid, **oln(blk.norm1))
    if amax_pool is not None:   # (a block whose launches are under the pair threshold: no slot for its dx)
        return (dx_in if need_dx else None), None
    return dx_in if need_dx else None


def patch_weight_planes(pe) -> torch.Tensor:
    """[1, D, C*P*P] bf16 plane of the patch-embedding weight (cached like the block weights: weight_planes)."""
    return weight_planes(pe.weight, 1).view(1, pe.weight.shape[0], -1)


PATCH_PAIRS = __import__("os").environ.get("TT_NO_PATCH_PAIRS") != "1"   # "f16x3": prepare_tokens on pair operands too (A/B aid: off)


def patch_weight_pairs(pe) -> torch.Tensor:
    """The conv weight of the patch embedding viewed [D, C P P] in fp16 pairs ("f16x3"), cached on the parameter like ``weight_planes``."""
    w = pe.weight
    hit = getattr(w, "_tt_planes", None)
    tag = _param_tag(w, 2)
    if hit is None or hit[0] != tag:
        hit = (tag, ops.split_pairs(w.detach().view(w.shape[0], -1)))
        w._tt_planes = hit
    return hit[1]


def patch_pairs_ok(vit, img: torch.Tensor) -> bool:
    """True when prepare_tokens runs on pair operands: the "f16x3" mode (above its row threshold), the pair GEMM's shape rules, and
    C P P <= 3 D (the im2col rows then fit tt_vit_forward's scratch) - tt_vit_forward's own rule, followed by the op-level path too."""
    pe = vit.patch_embed.proj
    D = pe.weight.shape[0]
    K = pe.weight[0].numel()
    return ops.pairs() and PATCH_PAIRS and K <= 3 * D and ops.patch_embed_pairs_ok(vit.patch_embed.patch_size, img.shape[-1], K, D)


def patch_planes_ok(vit, img: torch.Tensor) -> bool:
    """True when prepare_tokens runs on bf16 operands: the "bf16" mode, the plane GEMM's shape rules, and C P P <= 9 D (the im2col
    rows then fit tt_vit_forward's scratch) - tt_vit_forward's own rule, followed by the op-level path too so that both give the
    same bits."""
    pe = vit.patch_embed.proj
    D = pe.weight.shape[0]
    K = pe.weight[0].numel()
    return ops.plane_count() == 1 and K <= 9 * D and ops.patch_embed_planes_ok(vit.patch_embed.patch_size, img.shape[-1], K, D)


def vit_params(vit, first: int, last: int, pos: Optional[torch.Tensor] = None):
    """The parameter table ``ops.vit_forward`` (tt_vit_forward) reads, for blocks [first, last) of ``vit`` in the arithmetic the
    precision mode selects for blocks that keep nothing (fp32 operands, or the bf16 planes of ``weight_planes``).  Returns
    (struct, keep-alive list); rebuilt per call - parameters are re-homed by ``.to()`` / the EMA flattening, and the teacher's
    planes are re-split every step."""
    from . import _lib

    D = vit.patch_embed.proj.weight.shape[0]
    planes = ops.plane_count() if D % 64 == 0 and vit.blocks[0].mlp.fc1.weight.shape[0] % 64 == 0 else 0
    n = last - first
    arr = (_lib.VitBlockParams * max(n, 1))()
    keep = [arr]
    for j in range(n):
        blk = vit.blocks[first + j]
        at, mlp = blk.attn, blk.mlp
        b = arr[j]
        for name, t in (("norm1_w", blk.norm1.weight), ("norm1_b", blk.norm1.bias), ("qkv_w", at.qkv.weight), ("qkv_b", at.qkv.bias),
                        ("proj_w", at.proj.weight), ("proj_b", at.proj.bias), ("norm2_w", blk.norm2.weight), ("norm2_b", blk.norm2.bias),
                        ("fc1_w", mlp.fc1.weight), ("fc1_b", mlp.fc1.bias), ("fc2_w", mlp.fc2.weight), ("fc2_b", mlp.fc2.bias)):
            setattr(b, name, t.data_ptr() if t is not None else None)
        if planes:
            for name, w in (("qkv_wp", at.qkv.weight), ("proj_wp", at.proj.weight), ("fc1_wp", mlp.fc1.weight), ("fc2_wp", mlp.fc2.weight)):
                wp = weight_planes(w, planes)
                keep.append(wp)
                setattr(b, name, wp.data_ptr())
    pe = vit.patch_embed.proj
    vp = _lib.VitParams()
    vp.patch_w, vp.patch_b, vp.cls = pe.weight.data_ptr(), pe.bias.data_ptr(), vit.cls_token.data_ptr()
    vp.pos = pos.data_ptr() if pos is not None else None
    vp.blocks, vp.n_blocks = arr, n
    vp.norm_w, vp.norm_b = vit.norm.weight.data_ptr(), vit.norm.bias.data_ptr()
    vp.dim, vp.heads, vp.hidden, vp.patch, vp.planes = D, vit.num_heads, vit.blocks[0].mlp.fc1.weight.shape[0], vit.patch_embed.patch_size, planes
    vp.patch_wp = None
    if planes == 1 and pos is not None:      # (tt_vit_forward applies the shape rules of patch_planes_ok itself)
        wp = patch_weight_planes(pe)
        keep.append(wp)
        vp.patch_wp = wp.data_ptr()
    elif planes == 2 and PATCH_PAIRS and pos is not None and pe.weight[0].numel() % 32 == 0:   # (... and those of patch_pairs_ok)
        wp = patch_weight_pairs(pe)
        keep.append(wp)
        vp.patch_wp = wp.data_ptr()
    keep.append(pos)
    return vp, keep


# Two HIP streams inside a step (round 6; measured in round 5, tools/two_stream_probe.py: -3.7 % on the frozen blocks, -6.2 % on the trainable
# blocks' forward): the persistent GEMMs run one 8-wave workgroup per CU over whole rounds of tiles + a remainder round, and a launch's last
# round leaves CUs idle that the NEXT kernel of the same stream cannot use; an INDEPENDENT chain on a second stream can.  Two places have one:
#   * the frozen blocks (tt_vit_forward over blocks [0, first trainable)): the batch as two halves of frames, one per stream;
#   * the trainable blocks' forward: the frames that keep nothing (all but the target frames) and the kept target frames are separate chains
#     already (vit_tokens' lo / hi); lo goes to the side stream.
# Fork / join by stream waits on both sides of the section (every side-stream launch is bracketed by them: buffers allocated on one stream
# and used on the other need no record_stream).  A CAPTURED step (TimeT.enable_step_graph) stays on one stream: ``two_streams``.
# Off: TT_SINGLE_STREAM=1, and whenever launches are being timed one by one (``ops.fine_grained()``: bench.py's roofline pass - a kernel that
# shares the chip with another stream's kernel has no launch duration of its own).  Results: the frames of a batch are independent, so the two
# halves compute what the whole batch computes - bit for bit wherever a launch's grid decomposition does not change an accumulation order
# (the K-split of a K = 1536 launch's left-over tiles does: those rows agree to fp32 rounding; tests/test_hip_timet.py).
TWO_STREAMS = __import__("os").environ.get("TT_SINGLE_STREAM") != "1"
# Below 64 frames of ViT-S/16 (16 clips x 4) a launched step is host-bound and the forks only cost: 12 clips 4.85 ms with the streams against
# 3.96 on one, 8 clips 4.01 against 3.72; from 16 clips on they pay (4.52 against 4.68; 32 clips: -3 ... -8 %).
TWO_STREAMS_MIN_FRAMES = int(__import__("os").environ.get("TT_TWO_STREAMS_MIN_FRAMES", "64"))   # (the environment variable: sweeps only)
# the label propagation's similarities on the side stream, beside the Sinkhorn solve (ops.label_propagate_sims); "0": in the propagation's call (A/B)
LP_SIMS_ON_SIDE = __import__("os").environ.get("TT_LP_SIMS_SIDE", "1") != "0"
_SIDE_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}
# experiment (sweeps only): "p0:p1:p2:p3" = HIP stream priorities (0 default, -1 higher) of the side streams 0 .. 3 below; unset: 2 and 3 are
# stream 0 and every priority is the default.  Measured (round 6, C2, profiles/r06_step_knob_sweeps.txt): four side streams of equal priority
# 7.10 - 7.14 ms against 7.04 - 7.07 with two; the frames-that-keep-nothing chain and / or the teacher at -1: 7.08 - 7.11; all at -1: 7.05 - 7.10.
_SIDE_PRIORITIES = [int(v) for v in __import__("os").environ["TT_SIDE_PRIORITIES"].split(":")] if __import__("os").environ.get("TT_SIDE_PRIORITIES") else None


def side_stream(device, which: int = 0) -> "torch.cuda.Stream":
    """The side stream(s) of a device: 0 = the second chain of a section (see TWO_STREAMS), 1 = the EMA teacher's blocks and head; 2 = the
    trainable blocks' frames that keep nothing, 3 = the weight gradients (both are stream 0 unless TT_SIDE_PRIORITIES separates them)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if _SIDE_PRIORITIES is None and which >= 2:
        which = 0
    s = _SIDE_STREAMS.get((idx, which))
    if s is None:
        prio = _SIDE_PRIORITIES[which] if _SIDE_PRIORITIES is not None else 0
        s = _SIDE_STREAMS[(idx, which)] = torch.cuda.Stream(device=idx, priority=prio)
    return s


def two_streams(device, frames: int) -> bool:
    """Side streams for this section?  Not while launches are timed one by one, not below TWO_STREAMS_MIN_FRAMES frames, and NOT under a
    hipGraph capture: ROCm 7.2 replays a graph with forks and joins far slower than the launches it replaces (measured, round 6: C1 with
    the fork / join captured 6.2 ms against 1.95 on one captured stream; 8 clips 7.6 against 3.1) - a captured step stays on one stream."""
    return (TWO_STREAMS and torch.device(device).type == "cuda" and not ops.fine_grained() and frames >= TWO_STREAMS_MIN_FRAMES
            and not torch.cuda.is_current_stream_capturing())


def vit_tokens(vit, img: torch.Tensor, frame_map: Optional[torch.Tensor] = None, save_blocks: Optional[Dict[int, dict]] = None,
               last_block_probs: bool = False, last_block_aux: Optional[dict] = None, tap: Optional[dict] = None,
               save_from_frame: int = 0, on_tap=None):
    """prepare_tokens + all blocks (dino_vision_transformer.py:236-252).  Returns (tokens before the final norm, attention
    probabilities of the last block or None).  ``last_block_aux`` receives the last block's qkv.
    ``tap = {"block": i, "rows": r}`` receives under ``"x"`` a private copy of the first ``r`` frames' residual stream as it
    ENTERS block ``i`` (``i == depth``: as it leaves the last block) - what an EMA teacher sharing blocks [0, i) continues from;
    ``on_tap(tap)`` is called as soon as it exists (the teacher's own blocks can then start on another stream, beside the student's).

    ``save_blocks`` = {block id: dict} keeps the activations of those blocks for a later ``block_backward`` - of the frames
    [save_from_frame, F) only: a gradient reaches only the target frames (time_tuning.py:296-302), so from the first kept block
    on the pass runs as TWO streams, frames [0, save_from_frame) that keep nothing (in place; the bf16-plane kernels when a
    precision mode selects them) and the kept frames.  The tokens then come back as the pair (lo, hi); with
    ``save_from_frame == 0`` (or nothing kept) as one tensor."""
    pe = vit.patch_embed.proj
    D = pe.weight.shape[0]
    probs = None
    depth = len(vit.blocks)
    first_saved = min(save_blocks) if save_blocks else depth
    split = save_blocks and save_from_frame > 0
    lo = hi = None
    pos = vit.pos_table(img.shape[-2], img.shape[-1])
    done = 0
    if ops.fine_grained():
        if patch_planes_ok(vit, img):   # the same decision tt_vit_forward makes from VitParams.patch_wp
            x = ops.patch_embed_fwd_planes(img, patch_weight_planes(pe), pe.bias, vit.cls_token.view(D), pos, vit.patch_embed.patch_size, frame_map)
        elif patch_pairs_ok(vit, img) and ops.plane_count_for((img.shape[0] if frame_map is None else frame_map.numel()) * pos.shape[0]) == 2:
            x = ops.patch_embed_fwd_pairs(img, patch_weight_pairs(pe), pe.bias, vit.cls_token.view(D), pos, vit.patch_embed.patch_size, frame_map)
        else:
            x = ops.patch_embed_fwd(img, pe.weight.view(D, -1), pe.bias, vit.cls_token.view(D), pos, vit.patch_embed.patch_size, frame_map)
    else:
        # ONE call (tt_vit_forward) for prepare_tokens and every leading block that keeps nothing and is not tapped
        done = min(first_saved, tap["block"] if tap is not None else depth, depth - 1 if last_block_probs else depth)
        Fr = img.shape[0] if frame_map is None else frame_map.numel()
        P_ = vit.patch_embed.patch_size
        x = torch.empty((Fr, 1 + (img.shape[-2] // P_) * (img.shape[-1] // P_), D), dtype=f32, device=img.device)
        want_qkv = last_block_aux is not None and done == depth
        if not want_qkv and two_streams(img.device, Fr):
            # the batch as two halves of frames on two streams (see TWO_STREAMS): the first half on the side stream, the second on this one
            h = Fr // 2
            cur, side = torch.cuda.current_stream(), side_stream(img.device)
            fm = frame_map if frame_map is not None else None
            params = vit_params(vit, 0, done, pos)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.vit_forward(params, done, x[:h], img=img if fm is not None else img[:h], frame_map=None if fm is None else fm[:h])
            ops.vit_forward(params, done, x[h:], img=img if fm is not None else img[h:], frame_map=None if fm is None else fm[h:])
            cur.wait_stream(side)
        else:
            _, qkv_last, _ = ops.vit_forward(vit_params(vit, 0, done, pos), done, x, img=img, frame_map=frame_map, last_qkv=want_qkv)
            if want_qkv:
                last_block_aux["qkv"] = qkv_last
    for i, blk in enumerate(vit.blocks):
        if i < done:
            continue
        if tap is not None and tap["block"] == i:
            tap["x"] = (x if lo is None else lo)[: tap["rows"]].clone()
            if on_tap is not None:
                on_tap(tap)
        sv = save_blocks.get(i) if save_blocks is not None else None
        aux = last_block_aux if i == depth - 1 else None
        if split and i >= first_saved and lo is None and two_streams(img.device, x.shape[0]) and (tap is None or tap["block"] <= i or tap["block"] == depth):
            # blocks [i, depth) as two concurrent chains (see TWO_STREAMS): the frames that keep nothing on the side stream, the kept ones here
            lo, hi = x[:save_from_frame], x[save_from_frame:]
            want_aux = last_block_aux is not None
            aux_lo, aux_hi = ({} if want_aux else None), ({} if want_aux else None)
            cur, side = torch.cuda.current_stream(), side_stream(img.device, 2)
            # operands that are made lazily and cached (the pair / plane form of a weight at its first use) are made HERE, on this stream,
            # before the fork: made inside one chain they would be a cache hit for the other chain, on another stream, before they are written
            for rows in (lo.shape[0] * lo.shape[1], hi.shape[0] * hi.shape[1]):
                planes = ops.plane_count_for(rows)
                if planes and D % 64 == 0 and vit.blocks[i].mlp.fc1.weight.shape[0] % 64 == 0:
                    for j in range(i, depth):
                        b_ = vit.blocks[j]
                        for w_ in (b_.attn.qkv.weight, b_.attn.proj.weight, b_.mlp.fc1.weight, b_.mlp.fc2.weight):
                            weight_planes(w_, planes)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for j in range(i, depth):
                    lo = block_forward(lo, vit.blocks[j], vit.num_heads, None, aux_lo if j == depth - 1 else None)
            for j in range(i, depth):
                hi = block_forward(hi, vit.blocks[j], vit.num_heads, save_blocks.get(j), aux_hi if j == depth - 1 else None)
            cur.wait_stream(side)
            if want_aux:
                last_block_aux["qkv_lo"], last_block_aux["qkv_hi"] = aux_lo["qkv"], aux_hi["qkv"]
            break
        if split and i >= first_saved:
            if lo is None:
                lo, hi = x[:save_from_frame], x[save_from_frame:]
            aux_lo = {} if aux is not None else None
            aux_hi = {} if aux is not None else None
            lo = block_forward(lo, blk, vit.num_heads, None, aux_lo)
            hi = block_forward(hi, blk, vit.num_heads, sv, aux_hi)
            if aux is not None:
                aux["qkv_lo"], aux["qkv_hi"] = aux_lo["qkv"], aux_hi["qkv"]
            continue
        if last_block_probs and i == depth - 1:
            probs = last_block_attention(x, blk, vit.num_heads)
        x = block_forward(x, blk, vit.num_heads, sv, aux)
    if tap is not None and tap["block"] == depth:
        tap["x"] = (x if lo is None else lo)[: tap["rows"]].clone()
        if on_tap is not None:
            on_tap(tap)
    return (x if lo is None else (lo, hi)), probs


def vit_blocks(vit, x: torch.Tensor, first: int, last_block_aux: Optional[dict] = None) -> torch.Tensor:
    """Blocks [first, depth) of ``vit`` on a residual stream x [F,N,D] that is the caller's to overwrite."""
    depth = len(vit.blocks)
    if not ops.fine_grained() and first < depth:
        _, qkv_last, _ = ops.vit_forward(vit_params(vit, first, depth), depth - first, x, last_qkv=last_block_aux is not None)
        if last_block_aux is not None:
            last_block_aux["qkv"] = qkv_last
        return x
    for i in range(first, depth):
        x = block_forward(x, vit.blocks[i], vit.num_heads, None, last_block_aux if i == depth - 1 else None)
    return x


def last_block_attention(x: torch.Tensor, blk, num_heads: int) -> torch.Tensor:
    """Attention probabilities of a block (Block.forward(return_attention=True), dino_vision_transformer.py:147-150)."""
    Fr, N, D = x.shape
    h1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias)
    qkv = ops.linear_fwd(h1.view(Fr * N, D), blk.attn.qkv.weight, blk.attn.qkv.bias)
    _, _, probs = ops.attention_fwd(qkv.view(Fr, N, 3 * D), num_heads, return_probs=True)
    return probs


# ------------------------------------------------------------------------------------------------
# projection head (models.py:915-926): Linear GELU Linear GELU Linear GELU Linear
# ------------------------------------------------------------------------------------------------

def head_linears(head) -> List[torch.nn.Linear]:
    return [m for m in head if isinstance(m, torch.nn.Linear)]


def _head_pairs_ok(lins, rows: int) -> bool:
    return ops.plane_count_for(rows) == 2 and all(l.weight.shape[0] % 64 == 0 and l.weight.shape[1] % 64 == 0 and l.bias is not None for l in lins)


def head_forward(x: torch.Tensor, head, save: Optional[dict] = None) -> torch.Tensor:
    lins = head_linears(head)
    if _head_pairs_ok(lins, x.shape[0]):   # the "f16x3" mode: every layer on pair operands; the GELU epilogues write the next layer's operand
        xp = ops.split_pairs(x.contiguous())
        acts, pres = [xp], []
        for i, lin in enumerate(lins):
            last = i == len(lins) - 1
            o = ops.linear_fwd_pairs(xp, weight_planes(lin.weight, 2), lin.bias, act=0 if last else 1, out_f32=last, out_pairs=not last,
                                     save_pre=save is not None and not last)
            if last:
                x = o["y"]
            else:
                xp = o["pairs"]
                acts.append(xp)
                if save is not None:
                    pres.append(o["pre"])
        if save is not None:
            save["acts"], save["pres"], save["pairs"] = acts, pres, True
        return x
    if save is None and not ops.fine_grained():
        return ops.mlp_head_forward(x, [(lin.weight, lin.bias) for lin in lins])   # tt_mlp_head_forward: one call
    acts = [x]
    pres = []
    for i, lin in enumerate(lins):
        last = i == len(lins) - 1
        if save is not None and not last:
            x, pre = ops.linear_fwd(x, lin.weight, lin.bias, act=1, save_pre=True)
            pres.append(pre)
        else:
            x = ops.linear_fwd(x, lin.weight, lin.bias, act=0 if last else 1)
        acts.append(x)
    if save is not None:
        save["acts"], save["pres"] = acts, pres
    return x


def head_backward(dz: torch.Tensor, head, sv: dict, grads, out=None, dz_amax=None, amax_pool=None) -> torch.Tensor:
    """``dz_amax`` / ``amax_pool`` ("f16x3" mode): as ``block_backward`` - max |dz| from the kernel that wrote dz, and the slots in which
    each Linear's gelu' data gradient leaves the maximum of the next dy."""
    lins = head_linears(head)
    out = out or (lambda p: None)
    d, d_amax = dz, dz_amax
    for i in range(len(lins) - 1, -1, -1):
        lin = lins[i]
        if sv.get("pairs"):
            nxt = amax_pool.take() if (amax_pool is not None and i > 0) else None   # (i == 0: its dx feeds the final norm, not a Linear)
            d, grads[lin.weight], grads[lin.bias] = _bwd_both_pairs(d, lin.weight, sv["acts"][i], sv["pres"][i - 1] if i > 0 else None,
                                                                    dw_out=out(lin.weight), db_out=out(lin.bias), dy_amax=d_amax, dx_amax_out=nxt)
            d_amax = nxt
        else:
            d, grads[lin.weight], grads[lin.bias] = _bwd_both(d, lin.weight, sv["acts"][i], sv["pres"][i - 1] if i > 0 else None,
                                                              dw_out=out(lin.weight), db_out=out(lin.bias))
    return d


# ------------------------------------------------------------------------------------------------
# scores / assignment
# ------------------------------------------------------------------------------------------------

def prototype_scores(z: torch.Tensor, prototypes: torch.Tensor, save: Optional[dict] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """normalize(z) @ prototypes.T (time_tuning.py:130-141).  ``out``: a caller-owned [rows, K] destination (a row range of the
    batch + queue score matrix)."""
    if save is not None:
        zn, inv = ops.l2norm_fwd(z, save_inv=True)
        save["zn"], save["inv"] = zn, inv
    else:
        zn = ops.l2norm_fwd(z)
    return ops.linear_fwd(zn, prototypes, out=out)


def exchange_group():
    """``torch.distributed`` when the step has to exchange data, else None: an initialised process group with more than one
    rank - or with ONE rank when ``TT_EXCHANGE_SINGLE_RANK=1``, which issues the very same RCCL calls (async all-gather of the
    score rows, bucketed async all-reduce of the gradients) on a one-rank communicator: how the ``nccl`` code path is executed
    on a 1-GPU box, where RCCL refuses two ranks on one device."""
    import os

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size() > 1 or os.environ.get("TT_EXCHANGE_SINGLE_RANK") == "1":
        return dist
    return None


# Instrumented runs only (bench.py): a list that receives (kind, payload bytes, event before, event after) for every wait of the
# compute stream on a collective - the time between the two events is what the exchange EXPOSES (the stream had nothing else to do).
RCCL_PROFILE = None


def wait_collective(work, kind: str, nbytes: int, device) -> None:
    """``work.wait()`` (the compute stream waits for the collective's stream), bracketed by HIP events when ``RCCL_PROFILE`` is set."""
    if RCCL_PROFILE is None or not (isinstance(device, torch.device) and device.type == "cuda"):
        work.wait()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    work.wait()
    e1.record()
    RCCL_PROFILE.append((kind, int(nbytes), e0, e1))


_GATHER_BUFFERS: Dict[tuple, torch.Tensor] = {}


def _gather_buffer(rows: int, cols: int, device) -> torch.Tensor:
    """The all-gather's destination, kept across steps (one per shape and device: a training run has one or two - with / without the
    queue rows): the exchange allocates nothing per step.  Safe to reuse: the assignment that reads it is consumed (label propagation,
    cross entropy) before the next step's gather is enqueued on the same stream order."""
    key = (rows, cols, str(device))
    buf = _GATHER_BUFFERS.get(key)
    if buf is None:
        if len(_GATHER_BUFFERS) >= 8:
            _GATHER_BUFFERS.clear()
        buf = _GATHER_BUFFERS[key] = torch.empty((rows, cols), dtype=f32, device=device)
    return buf


# How a W > 1 solve exchanges: "allgather" (default: ONE all-gather of the score rows, every rank solves the global problem) or
# "allreduce" - the reference's own pattern (my_utils.py:250-272): the columns stay on their rank, the K row sums are all-reduced once
# per iteration (``iters`` small latency-bound collectives, a solve of B_loc instead of W B_loc columns).  ``--sinkhorn_exchange``.
SINKHORN_EXCHANGE = "allgather"
_SK_LOCAL: Dict[tuple, "ops.SinkhornLocal"] = {}
# Gradient exchange: 0 = a bucket per ``GradExchange.push`` (the four the fused backward produces, each hidden behind the backward that
# follows it), 1 = ONE all-reduce of all gradients at the end of the backward (fewer latency-bound collectives, nothing hidden).
GRAD_BUCKETS = 0
# What ``autotune_exchange`` measured and chose on THIS communicator (None: defaults / flags, nothing measured); bench.py prints it.
EXCHANGE_CHOICE: Optional[dict] = None


def autotune_exchange(run_step, device, reps: int = 3, margin: float = 0.97, log=None, state=None) -> Optional[dict]:
    """Which of the two Sinkhorn exchanges ("allgather": one 6.7 MB all-gather + a W-times-larger redundant solve; "allreduce": the
    reference's ``iters`` K-float all-reduces, my_utils.py:250-272) and which gradient exchange (four buckets behind the backward, or one
    at its end) is faster over THIS node's links is a property of the machine (xGMI ring latency against the solve's size) that no
    1-GPU box can tell: with more than one rank it is MEASURED, once, on the training step itself.  ``run_step()`` runs one forward +
    backward of the real workload (no parameter update); three configurations are timed in the same order on every rank - ``reps``
    steps each behind one warm-up step, barrier + synchronise on both sides, MAX over ranks so that every rank sees the same numbers and
    takes the same decision - and a variant replaces the default only when it is faster by more than ``1 - margin`` (3 %).
    Sets ``SINKHORN_EXCHANGE`` / ``GRAD_BUCKETS``, records the measurements in ``EXCHANGE_CHOICE`` and returns it; a no-op (None) without
    a process group of more than one rank.

    ``state = (snapshot, restore)`` (``TimeT.probe_state`` / ``restore_probe_state``): what a forward + backward of the workload changes
    besides gradients - the queue and its bookkeeping, the host generator the queue permutations are drawn from.  It is put back before
    EVERY timed configuration and at the end, so that all three are timed on the same workload (the queue fills while probing, and a full
    queue adds its rows to the Sinkhorn problem) and the run after the probe is seed for seed the run without one."""
    import time

    global SINKHORN_EXCHANGE, GRAD_BUCKETS, EXCHANGE_CHOICE
    dist = exchange_group()
    if dist is None or dist.get_world_size() < 2:
        return None
    dev = torch.device(device)
    cuda = dev.type == "cuda"

    snap = state[0]() if state is not None else None

    def timed(sk: str, buckets: int) -> float:
        global SINKHORN_EXCHANGE, GRAD_BUCKETS
        SINKHORN_EXCHANGE, GRAD_BUCKETS = sk, buckets
        if state is not None:
            state[1](snap)
        run_step()                                   # warm-up: buffers, communicator channels, the arena's layout for this bucket count
        if cuda:
            torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            run_step()
        if cuda:
            torch.cuda.synchronize(dev)
        t = torch.tensor([(time.perf_counter() - t0) * 1e3 / reps], dtype=torch.float64, device=dev if cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    keep = (SINKHORN_EXCHANGE, GRAD_BUCKETS)
    ms = {}
    try:
        ms["allgather/4"] = timed("allgather", 0)
        ms["allreduce/4"] = timed("allreduce", 0)
        sk = "allreduce" if ms["allreduce/4"] < margin * ms["allgather/4"] else "allgather"
        ms[f"{sk}/1"] = timed(sk, 1)
        buckets = 1 if ms[f"{sk}/1"] < margin * ms[f"{sk}/4"] else 0
    except Exception:
        SINKHORN_EXCHANGE, GRAD_BUCKETS = keep
        raise
    finally:
        if state is not None:
            state[1](snap)
    SINKHORN_EXCHANGE, GRAD_BUCKETS = sk, buckets
    EXCHANGE_CHOICE = dict(sinkhorn_exchange=sk, grad_buckets=4 if buckets == 0 else 1, ms_per_step=ms, reps=reps, world_size=dist.get_world_size(),
                           rule=f"a variant replaces the default (allgather, 4 buckets) when faster by > {100 * (1 - margin):.0f} %")
    if log is not None and dist.get_rank() == 0:
        log(f"exchange autotune (W = {dist.get_world_size()}): " + ", ".join(f"{k} {v:.3f} ms" for k, v in ms.items())
            + f" -> sinkhorn {sk}, {EXCHANGE_CHOICE['grad_buckets']} gradient bucket(s)")
    return EXCHANGE_CHOICE


def global_sinkhorn_allreduce(scores_local: torch.Tensor, rows_out: int, eps: float, iters: int, lib=None) -> torch.Tensor:
    """The reference's distributed Sinkhorn as it stands (my_utils.py:250-272): each rank keeps its own columns, ``iters`` all-reduces
    of the K row sums (the reference's two further collectives - the total mass, which cancels in the first row step, and a last row
    sum nobody reads - are not issued).  ``lib``: the CPU twin for the gloo tests (the HIP library otherwise)."""
    dist = exchange_group()
    W = dist.get_world_size() if dist is not None else 1
    local = scores_local.contiguous()
    B_loc, K = local.shape
    key = (B_loc, W, K, str(local.device), id(lib))
    sk = _SK_LOCAL.get(key)
    if sk is None:
        if len(_SK_LOCAL) >= 8:
            _SK_LOCAL.clear()
        sk = _SK_LOCAL[key] = ops.SinkhornLocal(B_loc, B_loc * W, K, local.device, lib=lib)
    u = sk.begin(local, eps)
    if iters <= 0:
        return sk.end(None, rows_out)
    for it in range(iters):
        if dist is not None:
            work = dist.all_reduce(u, async_op=True)
            wait_collective(work, "all_reduce(sinkhorn row sums)", u.numel() * 4, u.device)
        if it + 1 < iters:
            u = sk.step(u)
    return sk.end(u, rows_out)


def global_sinkhorn_begin(scores_local: torch.Tensor):
    """Starts the all-gather of the local score rows (asynchronous: RCCL moves them on its own stream while the caller
    keeps launching work that does not need the assignment) and returns the context ``global_sinkhorn_end`` consumes."""
    dist = exchange_group()
    if dist is None:
        return (scores_local, None, None)
    if SINKHORN_EXCHANGE == "allreduce":
        return (scores_local, None, "allreduce")   # nothing travels ahead: the row sums are exchanged inside the solve
    W = dist.get_world_size()
    local = scores_local.contiguous()
    gathered = _gather_buffer(W * local.shape[0], local.shape[1], local.device)
    try:
        work = dist.all_gather_into_tensor(gathered, local, async_op=True)
    except RuntimeError:  # backends without the flat all-gather (gloo on device tensors, used by the 1-GPU 2-rank test)
        work = dist.all_gather(list(gathered.chunk(W, dim=0)), local, async_op=True)
    return (local, gathered, work)


def global_sinkhorn_end(ctx, rows_out: int, eps: float, iters: int, solver=None) -> torch.Tensor:
    import torch.distributed as dist

    solver = ops.sinkhorn if solver is None else solver
    local, gathered, work = ctx
    if gathered is None and work == "allreduce":
        return global_sinkhorn_allreduce(local, rows_out, eps, iters)
    if gathered is None:
        return solver(local, iters, eps, row0=0, rows_out=rows_out)
    wait_collective(work, "all_gather(scores)", gathered.numel() * 4, gathered.device)
    return solver(gathered, iters, eps, row0=dist.get_rank() * local.shape[0], rows_out=rows_out)


def global_sinkhorn(scores_local: torch.Tensor, rows_out: int, eps: float, iters: int, solver=None) -> torch.Tensor:
    """find_optimal_assignment (time_tuning.py:157-168) with the reference's cross-rank semantics
    (my_utils.py:250-272): ONE all-gather of the local score rows over RCCL, then every rank solves the global
    K x (B_loc * W) problem and keeps the assignment of its own first ``rows_out`` rows.  Equal to the reference's
    1 + 1 + iters all-reduces; every rank must contribute the same number of rows, as there
    (``c = 1 / (B * world_size)``, my_utils.py:257).

    ``solver(scores, iters, eps, row0=, rows_out=)`` defaults to the HIP kernel; the CPU gloo tests inject the oracle
    to exercise the collective logic without a GPU."""
    return global_sinkhorn_end(global_sinkhorn_begin(scores_local), rows_out, eps, iters, solver)


class GradArena:
    """Persistent flat fp32 buffers that hold every gradient the exchange sends, laid out in the order the fused backward produces them,
    one contiguous range per bucket: the backward kernels write their dw / db straight into it (``GradExchange.out``), a bucket's
    all-reduce runs on a slice of it, and the averaged gradients are handed out as views - no ``torch.cat``, no allocation per step
    (what DDP's ``gradient_as_bucket_view`` does for the reference's wrapper, models.py:1295).  Built from the first step's buckets;
    rebuilt if a later step produces a different set or order of gradients (a changed ``requires_grad``).

    TWO buffers alternate from step to step (round 6): autograd adopts the views it is handed, so after a step ``.grad`` aliases that
    step's buffer - and in the reference's loop (``zero_grad()`` inside ``optimizer.step(loss)``, AFTER the next forward + backward has
    been launched, time_tuning.py:379-429) it still does while the next backward runs.  With one buffer every such gradient had to be
    given storage of its own before the buffer was overwritten (33 clone launches per step, 0.13 ms at C2); with two the next backward
    writes the OTHER buffer and nothing is copied.  Only a gradient that still aliases the buffer about to be written - a caller that
    accumulates over several backward calls without ``zero_grad()`` - is detached first (``detach_stale_grads``)."""

    def __init__(self):
        self.flats: List[torch.Tensor] = []                           # the two buffers
        self.cur = 0                                                  # the one the running / last backward writes
        self.all_slots: List[Dict[torch.nn.Parameter, torch.Tensor]] = []
        self.buckets: List[tuple] = []                                # (start, end, [params])

    ALIGN = 4   # floats: the backward kernels write dw / db through 16-byte stores (tt_linear_bwd_weight_pairs_tn rejects anything else)

    @property
    def flat(self) -> Optional[torch.Tensor]:
        return self.flats[self.cur] if self.flats else None

    @property
    def slots(self) -> Dict[torch.nn.Parameter, torch.Tensor]:       # param -> its view of the current buffer
        return self.all_slots[self.cur] if self.all_slots else {}

    def build(self, buckets: List[List[torch.nn.Parameter]], device) -> None:
        # every slot starts on a 16-byte boundary (a tensor whose numel is not a multiple of 4 - an odd bias, a custom head
        # width - misaligned every slot behind it); the padding floats stay zero and travel with their bucket, which is harmless
        pad = lambda n: (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = sum(pad(p.numel()) for keys in buckets for p in keys)
        self.flats = [torch.zeros((total,), dtype=f32, device=device) for _ in range(2)]
        self.all_slots, self.buckets, self.cur = [{}, {}], [], 0
        off = 0
        for keys in buckets:
            start = off
            for p in keys:
                for i in range(2):
                    self.all_slots[i][p] = self.flats[i][off:off + p.numel()].view(p.shape)
                off += pad(p.numel())
            self.buckets.append((start, off, list(keys)))

    def reset(self) -> None:
        self.flats, self.all_slots, self.buckets, self.cur = [], [], [], 0

    def next_step(self) -> None:
        """A new backward is about to write gradients: it takes the buffer the previous one did not use."""
        if self.flats:
            self.cur ^= 1
            self.detach_stale_grads()

    def detach_stale_grads(self) -> None:
        """A ``.grad`` that still aliases the buffer about to be written (it was adopted two backward calls ago and never reset: a caller
        that accumulates across steps) would be overwritten under the caller: give such a gradient its own storage first."""
        for p, view in self.slots.items():
            if p.grad is not None and p.grad.data_ptr() == view.data_ptr():
                p.grad = p.grad.clone()


class GradExchange:
    """The data-parallel gradient exchange, bucketed in the order the fused backward produces gradients (what DDP's
    bucketed all-reduce does in the reference, models.py:1295).  ``push(grads)`` takes the gradients that exist so far
    and are not yet sent and starts an asynchronous all-reduce (SUM) on their bucket - RCCL runs it on its own stream
    while the backward of the earlier blocks continues on the compute stream; ``finish(grads)`` sends the rest, waits for
    every bucket and returns the averaged gradients as views into the flat buffers.  Without an initialised process group
    (or with one rank) both are no-ops.  At C2 sizes the four buckets are prototypes + head (2.2 M floats), final norm +
    blocks.11 (1.8 M), the MLP half of blocks.10 (1.2 M, sent from inside that block's backward) and its attention half (0.6 M):
    only the last one is exposed.

    ``arena`` (a ``GradArena`` owned by the model): from the second step on the buckets are slices of ONE persistent buffer that the
    backward kernels write directly (``out``); the first step - and any step whose gradients differ from the recorded layout -
    flattens with ``torch.cat`` and records the layout."""

    def __init__(self, arena: Optional[GradArena] = None):
        self.dist = exchange_group()
        self.sent = set()
        self.buckets = []  # (keys, flat, work, in_arena)
        self.arena = arena if self.dist is not None else None
        self.use_arena = self.arena is not None and self.arena.flat is not None
        if self.use_arena:
            self.arena.next_step()
        self.recorded: List[List[torch.nn.Parameter]] = []

    def out(self, p) -> Optional[torch.Tensor]:
        """The destination a backward kernel should write ``p``'s gradient to (its slice of the flat bucket), or None."""
        if not self.use_arena or p is None:
            return None
        return self.arena.slots.get(p)

    def push(self, grads: Dict[torch.nn.Parameter, torch.Tensor], final: bool = False) -> None:
        if self.dist is None:
            return
        if GRAD_BUCKETS == 1 and not final:
            return                                   # one all-reduce of everything, issued by ``finish``
        keys = [k for k in grads if k not in self.sent and k.requires_grad]
        if not keys:
            return
        self.recorded.append(keys)
        flat = None
        if self.use_arena:
            i = len(self.buckets)
            same = i < len(self.arena.buckets) and len(self.arena.buckets[i][2]) == len(keys) and all(a is b for a, b in zip(self.arena.buckets[i][2], keys))
            if same:
                start, end, _ = self.arena.buckets[i]
                for k in keys:   # (a kernel that could not take a destination left its own tensor: one small copy, still no cat)
                    view = self.arena.slots[k]
                    if grads[k].data_ptr() != view.data_ptr():
                        view.copy_(grads[k].reshape(view.shape))
                flat = self.arena.flat[start:end]
            else:
                self.use_arena = False   # a different set / order of gradients: this step flattens by hand, the layout is rebuilt at finish
        in_arena = flat is not None
        # Weight gradients may have been launched on the side stream (``ops.wgrad_fork``): the bucket's all-reduce has to follow BOTH
        # streams.  Issued from the side stream behind a wait for this one, it does (the collective's stream waits for the stream it is
        # issued from) WITHOUT joining the compute stream at every bucket boundary: the data-gradient chain runs on.
        side = ops.wgrad_side()
        if flat is None:
            if side is not None:
                ops.wgrad_join(final=False)          # (the hand-flattened first step reads the gradients on this stream)
                side = None
            flat = torch.cat([grads[k].reshape(-1) for k in keys])
        if side is not None and flat.is_cuda:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                work = self.dist.all_reduce(flat, async_op=True)
        else:
            work = self.dist.all_reduce(flat, async_op=True)
        self.sent.update(keys)
        self.buckets.append((keys, flat, work, in_arena))

    def prescale_(self, root_grad: torch.Tensor) -> bool:
        """Applies the 1 / world_size of the mean at the ROOT of the backward (the loss gradient, one small tensor) instead of to
        every bucket after its all-reduce: every gradient is linear in it, so the summed buckets then ARE the means (bit for bit
        when the world size is a power of two).  Returns True when applied; ``finish`` must then be told ``scale=False``."""
        if self.dist is None or not root_grad.is_cuda:
            return False
        inv = torch.full((1,), 1.0 / self.dist.get_world_size(), dtype=f32, device=root_grad.device)
        ops.scale_tensors_([root_grad], inv)
        return True

    def finish(self, grads: Dict[torch.nn.Parameter, torch.Tensor], scale: bool = True) -> Dict[torch.nn.Parameter, torch.Tensor]:
        if self.dist is None:
            return grads
        self.push(grads, final=True)
        inv = torch.full((1,), 1.0 / self.dist.get_world_size(), dtype=f32, device=self.buckets[0][1].device) if self.buckets and scale else None
        for b_i, (keys, flat, work, in_arena) in enumerate(self.buckets):
            wait_collective(work, f"all_reduce(grad bucket {b_i})", flat.numel() * 4, flat.device)
            if not scale:
                pass                                 # (already the mean: prescale_)
            elif flat.is_cuda:
                ops.scale_tensors_([flat], inv)      # SUM -> mean
            else:
                flat *= inv                          # (CPU tensors: only the gloo tests of the collective logic)
            off = 0
            for k in keys:
                # (an arena bucket: the parameter's own 16-byte aligned slot - the slots are padded; a hand-flattened one: back to back)
                # (a FRESH view object either way: autograd adopts a gradient it holds the only reference to, and clones one it does not)
                grads[k] = self.arena.slots[k].view(k.shape) if in_arena else flat[off:off + k.numel()].view(k.shape)
                off += k.numel()
        if self.arena is not None and not self.use_arena and self.buckets and self.buckets[0][1].is_cuda:
            self.arena.build(self.recorded, self.buckets[0][1].device)   # the layout the next steps write into
        self.buckets = []
        return grads
