"""Per-op parity of the fp16-PAIR kernels - the fp32-accurate split mode "f16x3" (gemm_pairs8.hip, the PAIR instances of gemm_planes_kernel,
attention_pairs.hip, the pair-writing LayerNorm / split / transpose kernels) - through the C ABI.

Held to the SAME 2e-5 max-normalised bound as the f32-MFMA kernels (tests/test_hip_ops.py) against fp64 of the fp32 operands, to a relative-L2
bound beside it, AND to VERDICT r3's rule for calling a split mode f32-class: its error against fp64 must not exceed the f32-MFMA kernel's own
on the same operands."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, rel_l2
from timetuning_amd import synth

pytestmark = pytest.mark.gpu
TOL_F32 = 2e-5
TOL_L2 = 2e-6


def rnd(name, *shape, scale=1.0):
    return torch.from_numpy(synth.normal("pairs." + name, shape, scale))


def join(p):
    """fp16 pairs [..., 2 C] -> fp64 [..., C] on the host: hi + lo / 2^11."""
    q = p.double().view(*p.shape[:-1], -1, 2, 32)
    return (q[..., 0, :] + q[..., 1, :] / 2048.0).reshape(*p.shape[:-1], -1)


def test_split_pairs_layout_and_accuracy():
    from timetuning_amd import hip_ops as ops

    x = rnd("split", 64, 192, scale=3.0)
    x[0, :8] = torch.tensor([0.0, 1e-30, -65504.0, 6.0e-5, 1e-6, 3.3e-8, -2.5, 1024.0009765625])
    p = ops.split_pairs(x.cuda()).cpu()
    assert p.shape == (64, 384) and p.dtype == torch.float16
    g = p.view(64, 6, 2, 32)
    assert torch.equal(g[:, :, 0, :].reshape(64, 192), x.to(torch.float16))                        # hi = fp16(x), groups of 32 columns
    lo_ref = ((x - x.to(torch.float16).float()) * 2048.0).to(torch.float16)
    assert torch.equal(g[:, :, 1, :].reshape(64, 192), lo_ref)                                     # lo = fp16((x - hi) 2^11)
    err = (join(p) - x.double()).abs()
    assert (err <= 2.0 ** -23 * x.double().abs() + 2.0 ** -36).all()                               # 23 significant bits; an absolute floor far below fp32's
    assert torch.equal(ops.join_pairs(p.cuda()).cpu().double(), join(p).float().double())


def test_layernorm_pairs():
    from timetuning_amd import hip_ops as ops

    for D in (384, 96):   # the vectorised kernel and the general one
        x, g, b = rnd(f"ln.x{D}", 3, 197, D, scale=2.0), 1.0 + 0.1 * rnd(f"ln.g{D}", D), 0.1 * rnd(f"ln.b{D}", D)
        ref = F.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6)
        y, mean, rstd = ops.layernorm_fwd_pairs(x.cuda(), g.cuda(), b.cuda(), save_stats=True)
        assert y.shape == (591, 2 * D)
        assert rel_err(join(y.cpu()).view(3, 197, D), ref) < TOL_F32 and rel_l2(join(y.cpu()).view(3, 197, D), ref) < TOL_L2
        assert rel_err(mean.cpu(), x.double().mean(-1).view(-1)) < 1e-5
        yd = ops.layernorm_fwd_pairs(x.cuda(), g.cuda(), b.cuda(), drop_first_token=True)
        assert yd.shape == (3 * 196, 2 * D) and rel_err(join(yd.cpu()).view(3, 196, D), ref[:, 1:]) < TOL_F32


# persistent kernel (whole 256 x 128 tiles, K % 96 == 0, a grid that fills the chip) and the general one (ragged M, 64-wide N, K % 32)
SHAPES = [(25216, 1152, 384), (25216, 384, 1536), (8192, 1536, 384), (6304, 384, 384), (788, 384, 384), (1000, 1152, 384), (6272, 256, 512),
          (394, 64, 1536), (2048, 768, 3072), (300, 128, 96), (33000, 128, 96)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_pairs(M, N, K):
    from timetuning_amd import hip_ops as ops

    x, w, b = rnd(f"lin.x{M}.{K}", M, K), rnd(f"lin.w{N}.{K}", N, K, scale=0.05), rnd(f"lin.b{N}", N, scale=0.1)
    xp, wp = ops.split_pairs(x.cuda()), ops.split_pairs(w.cuda())
    y = ops.linear_fwd_pairs(xp, wp, b.cuda())["y"].cpu()
    ref = x.double() @ w.double().t() + b.double()
    e_pair, l_pair = rel_err(y, ref), rel_l2(y, ref)
    assert e_pair < TOL_F32 and l_pair < TOL_L2
    # the rule for calling the mode f32-class: not worse than the exact-f32 MFMA kernel on the same operands (5 % slack for the max, which
    # is one element's luck; none for the L2 norm)
    y32 = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda()).cpu()
    assert l_pair <= rel_l2(y32, ref) and e_pair <= 1.05 * rel_err(y32, ref)
    # a race in the counted-vmcnt schedule shows as a run-to-run difference
    for _ in range(3):
        assert torch.equal(ops.linear_fwd_pairs(xp, wp, b.cuda())["y"].cpu(), y)


@pytest.mark.parametrize("M,N,K", [(25216, 384, 384), (6304, 1536, 384), (591, 256, 128)])
def test_linear_pairs_epilogues(M, N, K):
    """bias + GELU + pre-activation + pair output; residual add in place; pair-only output - on the persistent and the general kernel."""
    from timetuning_amd import hip_ops as ops

    x, w, b, res = rnd(f"epi.x{M}", M, K), rnd(f"epi.w{N}", N, K, scale=0.1), rnd(f"epi.b{N}", N, scale=0.1), rnd(f"epi.r{M}.{N}", M, N)
    xp, wp = ops.split_pairs(x.cuda()), ops.split_pairs(w.cuda())
    pre_ref = x.double() @ w.double().t() + b.double()
    o = ops.linear_fwd_pairs(xp, wp, b.cuda(), act=1, out_pairs=True, save_pre=True)   # (pre_out: the general kernel)
    assert rel_err(o["pre"].cpu(), pre_ref) < TOL_F32 and rel_err(o["y"].cpu(), F.gelu(pre_ref)) < TOL_F32
    got = join(o["pairs"].cpu())
    assert rel_err(got, o["y"].cpu().double()) <= 2.0 ** -22          # the pairs ARE the fp32 result, split
    o1 = ops.linear_fwd_pairs(xp, wp, b.cuda(), act=1, out_f32=False, out_pairs=True)   # (the persistent kernel where the shape allows)
    assert o1["y"] is None and rel_err(join(o1["pairs"].cpu()), F.gelu(pre_ref)) < TOL_F32
    o2 = ops.linear_fwd_pairs(xp, wp, b.cuda(), out_f32=False, out_pairs=True)
    assert rel_err(join(o2["pairs"].cpu()), pre_ref) < TOL_F32
    rc = res.clone().cuda()
    o3 = ops.linear_fwd_pairs(xp, wp, b.cuda(), residual=rc, out=rc)
    assert o3["y"].data_ptr() == rc.data_ptr() and rel_err(rc.cpu(), pre_ref + res.double()) < TOL_F32
    o4 = ops.linear_fwd_pairs(xp, wp, None)
    assert rel_err(o4["y"].cpu(), pre_ref - b.double()) < TOL_F32
    # what a forward that keeps its backward's operands asks for: the fp32 pre-activation + GELU in pairs; fp32 y + the same in pairs
    o5 = ops.linear_fwd_pairs(xp, wp, b.cuda(), act=1, out_f32=False, out_pairs=True, save_pre=True)
    assert o5["y"] is None and rel_err(o5["pre"].cpu(), pre_ref) < TOL_F32 and rel_err(join(o5["pairs"].cpu()), F.gelu(pre_ref)) < TOL_F32
    assert rel_err(join(o5["pairs"].cpu()), F.gelu(o5["pre"].cpu().double())) < 2e-6     # the pairs are GELU of THAT pre-activation
    o6 = ops.linear_fwd_pairs(xp, wp, b.cuda(), out_pairs=True)
    assert rel_err(o6["y"].cpu(), pre_ref) < TOL_F32 and rel_err(join(o6["pairs"].cpu()), o6["y"].cpu().double()) <= 2.0 ** -22


@pytest.mark.parametrize("M,N,K", [(25216, 384, 384), (12608, 1152, 384), (25216, 1536, 384), (8100, 1024, 768), (70000, 128, 64)])
def test_four_wave_pair_gemm_experiment(M, N, K):
    """gemm_pairs4_kernel (round 6, knob TT_Q4, default OFF: DESIGN 5.1) - 128 x 128 tiles in four-wave workgroups, two per CU, the left-over
    tiles as half items in a second launch: every epilogue against fp64 at the pair kernels' bounds, bit for bit equal to the 8-wave kernel
    wherever that one does not K-split (same products in the same order per output element), run-to-run bit equality; ragged last tiles."""
    from timetuning_amd import hip_ops as ops

    x, w, b, res = rnd(f"q4.x{M}.{K}", M, K), rnd(f"q4.w{N}.{K}", N, K, scale=0.1), rnd(f"q4.b{N}", N, scale=0.1), rnd(f"q4.r{M}.{N}", M, N)
    xp, wp = ops.split_pairs(x.cuda()), ops.split_pairs(w.cuda())
    pre_ref = x.double() @ w.double().t() + b.double()

    def run_all():
        out = {}
        out["f32"] = ops.linear_fwd_pairs(xp, wp, b.cuda())["y"]
        rc = res.clone().cuda()
        ops.linear_fwd_pairs(xp, wp, b.cuda(), residual=rc, out=rc)
        out["res"] = rc
        out["pair"] = ops.linear_fwd_pairs(xp, wp, b.cuda(), out_f32=False, out_pairs=True)["pairs"]
        out["gelu"] = ops.linear_fwd_pairs(xp, wp, b.cuda(), act=1, out_f32=False, out_pairs=True)["pairs"]
        o = ops.linear_fwd_pairs(xp, wp, b.cuda(), out_pairs=True)
        out["both_y"], out["both_p"] = o["y"], o["pairs"]
        o = ops.linear_fwd_pairs(xp, wp, b.cuda(), act=1, out_f32=False, out_pairs=True, save_pre=True)
        out["bg_pre"], out["bg_p"] = o["pre"], o["pairs"]
        return out

    ops.set_tuning_knob("TT_Q8_KSPLIT", 0)          # (the 8-wave kernel's K-split changes an accumulation order: off for the bit comparison)
    try:
        base = run_all()
        ops.set_tuning_knob("TT_Q4", 1)
        q4 = run_all()
        again = run_all()
    finally:
        ops.set_tuning_knob("TT_Q4", 0)
        ops.set_tuning_knob("TT_Q8_KSPLIT", 1)
    for k_ in q4:
        assert torch.equal(q4[k_], again[k_]), k_
        assert torch.equal(q4[k_], base[k_]), k_
    assert rel_err(q4["f32"].cpu(), pre_ref) < TOL_F32 and rel_l2(q4["f32"].cpu(), pre_ref) < TOL_L2
    assert rel_err(q4["res"].cpu(), pre_ref + res.double()) < TOL_F32
    assert rel_err(join(q4["gelu"].cpu()), F.gelu(pre_ref)) < TOL_F32
    assert rel_err(join(q4["bg_p"].cpu()), F.gelu(q4["bg_pre"].cpu().double())) < 2e-6


@pytest.mark.parametrize("Fr,N,H,flash", [(3, 197, 6, 0), (2, 50, 2, 0), (1, 256, 12, 0), (2, 225, 3, 0), (5, 17, 1, 0),
                                          # the persistent loop of the resident kernel (round 5): more (frame, head) items than CUs, unevenly
                                          # (300 and 258 items on 256 workgroups: one or two items each; 768: three each)
                                          (50, 197, 6, 0), (43, 256, 6, 0), (128, 197, 6, 0),
                                          # the KV-tiled kernel: beyond 256 tokens (785 = C5's 448 x 448 frames: 7 + 6 + 6 + 6 query tiles; ragged
                                          # last stages; one stage + one key), and forced at the sizes of the resident kernel
                                          (2, 785, 6, 0), (2, 300, 2, 0), (1, 257, 1, 0), (1, 1030, 2, 0), (1, 513, 3, 0),
                                          (3, 197, 6, 1), (5, 17, 1, 1), (2, 128, 2, 1), (2, 129, 2, 1), (1, 256, 3, 1), (2, 65, 1, 1)])
def test_attention_pairs(Fr, N, H, flash):
    from timetuning_amd import hip_ops as ops

    ops.set_tuning_knob("TT_ATTN_PAIRS_FLASH", flash)
    try:
        _attention_pairs_case(ops, Fr, N, H)
    finally:
        ops.set_tuning_knob("TT_ATTN_PAIRS_FLASH", 0)


def _attention_pairs_case(ops, Fr, N, H):

    D = 64 * H
    qkv = rnd(f"att.{Fr}.{N}.{H}", Fr, N, 3 * D) * 1.5
    qkvp = ops.split_pairs(qkv.view(Fr * N, 3 * D).cuda()).view(Fr, N, 6 * D)
    q, k, v = qkv.double().view(Fr, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) * 64 ** -0.5
    ref = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(Fr, N, D)
    op, of, lse = ops.attention_fwd_pairs(qkvp, H, out_pairs=True, out_f32=True, save_lse=True)
    assert rel_err(of.cpu(), ref) < TOL_F32 and rel_l2(of.cpu(), ref) < TOL_L2
    assert rel_err(join(op.cpu()), ref) < TOL_F32
    assert rel_err(lse.cpu(), torch.logsumexp(sc, dim=-1)) < 1e-5
    o32, _, _ = ops.attention_fwd(qkv.cuda(), H)
    assert rel_l2(of.cpu(), ref) <= rel_l2(o32.cpu(), ref)          # not worse than the f32-MFMA attention kernel
    only_pairs = ops.attention_fwd_pairs(qkvp, H)[0]
    assert torch.equal(only_pairs, op)
    if N <= 256:   # one workgroup per item (the round-4 launch shape of the same kernel): the same bits
        ops.set_tuning_knob("TT_ATTN_PAIRS_PERSIST", 0)
        try:
            assert torch.equal(ops.attention_fwd_pairs(qkvp, H)[0], op)
        finally:
            ops.set_tuning_knob("TT_ATTN_PAIRS_PERSIST", 1)


@pytest.mark.parametrize("tn", [True, False])
@pytest.mark.parametrize("mag", [1e-3, 3e-8])
@pytest.mark.parametrize("M,N,K", [(6304, 1536, 384), (6304, 384, 1152), (591, 256, 512), (3152, 768, 3072), (6299, 384, 384), (45, 128, 256)])
def test_backward_products_on_pairs(M, N, K, tn, mag):
    """dx = dy @ w (* gelu'(pre)), dw = dy^T @ x and db = dy.sum(0) of an nn.Linear on pair operands (the "f16x3" mode's backward): against
    fp64 at the f32 bound, not worse than the f32-MFMA backward kernels, split-K fold included.  tn: the weight gradient from ROW pairs
    (gemm_pairs_tn.hip: ragged M read as zeros) / from transposed pairs (the route of the shapes that kernel does not take)."""
    from timetuning_amd import engine, hip_ops as ops

    # mag: the gradient's magnitude - 3e-8 is what a C2 step's deepest dy tensors really look like (tools/grad_err.py: medians 2e-8 .. 3e-6,
    # every element below fp16's smallest normal 6.1e-5); the split scales a gradient by a power of two first (tt_split_pairs_dual scale_out)
    dy, w, x = rnd(f"bwd.dy{M}.{N}", M, N, scale=mag), rnd(f"bwd.w{N}.{K}", N, K, scale=0.05), rnd(f"bwd.x{M}.{K}", M, K)
    pre = rnd(f"bwd.pre{M}.{K}", M, K)
    xp = ops.split_pairs(x.cuda())
    assert ops.TN_WGRAD
    ops.TN_WGRAD = tn
    try:
        assert ops.bwd_weight_pairs_tn_ok(M, N, K) == (tn and N % 128 == 0 and K % 128 == 0)
        dx, dw, db = engine._bwd_both_pairs(dy.cuda(), w.cuda(), xp, pre.cuda())
        if tn and ops.bwd_weight_pairs_tn_ok(M, N, K):   # run-to-run bit equality (fixed fold order)
            assert torch.equal(engine._bwd_both_pairs(dy.cuda(), w.cuda(), xp, pre.cuda())[1], dw)
    finally:
        ops.TN_WGRAD = True
    from oracle import timet_oracle as O  # noqa: F401  (the checker's GELU' is torch's own below)

    pd = pre.double().requires_grad_(True)
    F.gelu(pd).sum().backward()
    dx_ref = (dy.double() @ w.double()) * pd.grad
    dw_ref = dy.double().t() @ x.double()
    assert rel_err(dx.cpu(), dx_ref) < TOL_F32 and rel_l2(dx.cpu(), dx_ref) < TOL_L2
    assert rel_err(dw.cpu(), dw_ref) < TOL_F32 and rel_l2(dw.cpu(), dw_ref) < TOL_L2
    assert rel_err(db.cpu(), dy.double().sum(0)) < TOL_F32
    dx32, dw32, _ = ops.linear_bwd(dy.cuda(), w.cuda(), x.cuda(), gelu_pre=pre.cuda())
    assert rel_l2(dw.cpu(), dw_ref) <= rel_l2(dw32.cpu(), dw_ref) and rel_l2(dx.cpu(), dx_ref) <= 1.05 * rel_l2(dx32.cpu(), dx_ref)


def test_transposed_pairs():
    from timetuning_amd import hip_ops as ops

    x = rnd("tr.x", 197, 96, scale=2.0)
    t, row, sums = ops.split_pairs_dual(x.cuda(), want_row=True, want_colsum=True)
    assert t.shape == (96, 2 * 224) and torch.equal(row, ops.split_pairs(x.cuda()))
    xt = torch.zeros(96, 224)
    xt[:, :197] = x.t()
    assert torch.equal(t.cpu(), ops.split_pairs(xt.cuda()).cpu())                    # zero beyond R, same split
    assert rel_err(sums.cpu(), x.double().sum(0)) < 1e-6
    assert torch.equal(ops.transpose_pairs(row).cpu(), t.cpu())                      # the 16-bit transpose of a pair tensor gives the same bits
    t2, _, _ = ops.split_pairs_dual(x.cuda(), rpad=256)
    assert t2.shape == (96, 512) and torch.equal(t2[:, :448].cpu().view(96, 7, 64), t.cpu().view(96, 7, 64)) and not t2[:, 448:].any()


@pytest.mark.parametrize("M,N,K", [(6304, 1536, 384), (6299, 384, 1152), (591, 256, 512), (45, 128, 256)])
@pytest.mark.parametrize("scaled", [False, True])
def test_bias_gradient_fold_rides_on_the_weight_gradient_fold(M, N, K, scaled):
    """tt_split_pairs_dual_parts leaves the column partials of a dy unfolded and tt_linear_bwd_weight_pairs_tn_bias folds them in the launch
    that folds the weight gradient's split partials (round 5: one launch less per Linear): dw and db bit for bit as the two-launch route."""
    from timetuning_amd import hip_ops as ops

    dy = (rnd(f"cf.dy.{M}.{N}", M, N) * (1e-5 if scaled else 1.0)).cuda()
    xp = ops.split_pairs(rnd(f"cf.x.{M}.{K}", M, K).cuda())
    a = ops.split_pairs_dual(dy, want_row=True, want_colsum=True, want_t=False, scaled=scaled)
    dw_a = ops.linear_bwd_weight_pairs_tn(a[1], xp, dy_scale=a[3] if scaled else None)
    b = ops.split_pairs_dual(dy, want_row=True, want_colsum=True, want_t=False, scaled=scaled, colsum_parts=True)
    assert torch.equal(a[1], b[1]) and tuple(b[2].shape) == (((M + 31) // 32 * 32 + 63) // 64, N)
    dw_b, db_b = ops.linear_bwd_weight_pairs_tn(b[1], xp, dy_scale=b[3] if scaled else None, colsum_parts=b[2])
    assert torch.equal(dw_a, dw_b) and torch.equal(a[2], db_b)
    assert rel_err(db_b.cpu(), dy.double().sum(0).cpu()) < 1e-6


@pytest.mark.parametrize("M,N,K,gelu", [(6304, 1536, 384, True), (6304, 384, 1536, False), (788, 384, 384, True), (6272, 1024, 512, True)])
def test_gradient_scale_from_the_producers_maximum(M, N, K, gelu):
    """Round 5: the kernel that PRODUCES a dy publishes max |dy| (``amax_out``) and the scaled pair split of that dy takes it (``dy_amax``)
    instead of making a max pass of its own.  The published maximum is exact, so the power-of-two scale - and with it every bit of dx, dw
    and db - is the one the max pass finds; the data-gradient kernels (persistent x gelu' epilogue, general kernel) publish max |dx|."""
    from timetuning_amd import engine, hip_ops as ops

    dy = (rnd(f"am.dy.{M}.{N}", M, N) * 3e-6).cuda()
    w = (rnd(f"am.w.{N}.{K}", N, K) * 0.05).cuda()
    xp = ops.split_pairs(rnd(f"am.x.{M}.{K}", M, K).cuda())
    pre = rnd(f"am.pre.{M}.{K}", M, K).cuda() if gelu else None
    ref = engine._bwd_both_pairs(dy, w, xp, pre)
    pool = ops.AmaxPool.get(dy.device)
    pool.reset()
    a_dy, a_dx = pool.take(), pool.take()
    a_dy[0] = dy.abs().max()                     # (what layernorm_bwd / attention_bwd / ... leave in one of the slot's 16 ways: test_cpu_twin)
    # (max |dx| is asked of the x gelu' data gradients only - the dx that are a next Linear's dy; without it the request would route
    # the product to the general kernel)
    got = engine._bwd_both_pairs(dy, w, xp, pre, dy_amax=a_dy, dx_amax_out=a_dx if gelu else None)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    if gelu:
        assert a_dx.max().item() == got[0].abs().max().item()
    # a stale (too small) maximum would overflow the split: the range flag says so
    ops.range_flag(dy.device).zero_()
    a_dy.fill_(float(dy.abs().max()) * 2.0 ** -6)
    engine._bwd_both_pairs(dy, w, xp, pre, dy_amax=a_dy)
    assert int(ops.range_flag(dy.device)[0].item()) == 1
    ops.range_flag(dy.device).zero_()


def test_hip_pair_ops_equal_their_cpu_twins():
    """The HIP library and the plain-C twins (oracle/tt_cpu.c) through one call site with identical prototypes."""
    from oracle import cpu_twin
    from timetuning_amd import _lib

    hip, twin = _lib.load(), cpu_twin.load()
    st = torch.cuda.current_stream().cuda_stream
    M, N, K = 70, 64, 96
    x, w, b = rnd("tw.x", M, K), rnd("tw.w", N, K, scale=0.1), rnd("tw.b", N, scale=0.1)

    def run(lib, pre, dev):
        to = (lambda t: t.cuda()) if dev else (lambda t: t.clone())
        ptr = lambda t: t.data_ptr()
        xs, ws, bs = to(x), to(w), to(b)
        xp, wp = torch.empty(M, 2 * K, dtype=torch.float16, device=xs.device), torch.empty(N, 2 * K, dtype=torch.float16, device=xs.device)
        f = lambda name: getattr(lib, pre + name)
        flag = torch.zeros(4, dtype=torch.int32, device=xs.device)   # (the range flag: a device int here, a host int for the twin)
        assert f("split_pairs")(ptr(xs), ptr(xp), M * K, ptr(flag), st if dev else None) == 0
        assert f("split_pairs")(ptr(ws), ptr(wp), N * K, ptr(flag), st if dev else None) == 0
        y = torch.empty(M, N, device=xs.device); yp = torch.empty(M, 2 * N, dtype=torch.float16, device=xs.device); pr = torch.empty(M, N, device=xs.device)
        assert f("linear_fwd_pairs")(ptr(xp), ptr(wp), ptr(bs), None, ptr(y), ptr(pr), ptr(yp), M, N, K, 1, None, 0, ptr(flag), st if dev else None) == 0
        t = torch.empty(K, 2 * 96, dtype=torch.float16, device=xs.device)
        assert f("transpose_pairs")(ptr(xp), ptr(t), M, K, 96, st if dev else None) == 0
        # ... and an operand beyond fp16's range raises the flag on both sides
        big = xs.clone(); big[3, 5] = 7.0e4
        assert f("split_pairs")(ptr(big), ptr(torch.empty_like(xp)), M * K, ptr(flag[1:]), st if dev else None) == 0
        if dev:
            torch.cuda.synchronize()
        assert flag.cpu().tolist() == [0, 1, 0, 0]
        return [v.cpu() for v in (xp, wp, y, yp, pr, t)]

    a, c = run(hip, "tt_", True), run(twin, "tt_cpu_", False)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and torch.equal(a[5], c[5])          # conversions and transposes: bit for bit
    assert rel_err(a[2], c[2]) < TOL_F32 and rel_err(a[4], c[4]) < TOL_F32 and rel_err(join(a[3]), join(c[3])) < TOL_F32


def test_pairs8_load_part_orders_give_the_same_bits():
    """TT_Q8_ORDER only moves a wave's DMA instructions relative to its fragment reads inside a phase of gemm_pairs8_kernel: every order
    must leave the same bits (a hazard in the counted-vmcnt schedule would show here or as a run-to-run difference)."""
    from timetuning_amd import hip_ops as ops

    M, N, K = 25216, 384, 384
    xp, wp = ops.split_pairs(rnd("ord.x", M, K).cuda()), ops.split_pairs(rnd("ord.w", N, K, scale=0.05).cuda())
    b = rnd("ord.b", N).cuda()
    outs = []
    try:
        for order in (3, 0, 1, 2):
            ops.set_tuning_knob("TT_Q8_ORDER", order)
            outs.append(ops.linear_fwd_pairs(xp, wp, b, act=1, out_f32=False, out_pairs=True)["pairs"].clone())
    finally:
        ops.set_tuning_knob("TT_Q8_ORDER", 3)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])


def test_batched_weight_operand_refresh():
    """engine.refresh_pair_operands: the row pairs and transposed pairs of every stale weight in ONE launch, written into the existing
    buffers, equal to the lazy per-weight conversions bit for bit; operands that were never made are not made."""
    from timetuning_amd import engine, hip_ops as ops

    shapes = [(384, 384), (1152, 384), (1536, 384), (384, 1536), (200, 256), (65, 96)]
    ws = [torch.nn.Parameter(rnd(f"refresh.w{i}", n, k, scale=0.05).cuda()) for i, (n, k) in enumerate(shapes)]
    ws[4].requires_grad_(False)
    never = torch.nn.Parameter(rnd("refresh.never", 128, 64).cuda())
    rows = [engine.weight_planes(w, 2) for w in ws]
    ts = [engine.weight_pairs_t(w) if w.requires_grad else None for w in ws]
    assert engine.refresh_pair_operands(ws + [never]) == 0                      # nothing stale
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5).add_(0.01)                                               # (advances the version counter: every tag is stale)
    assert engine.refresh_pair_operands(ws + [never]) == len(ws)
    assert getattr(never, "_tt_planes", None) is None and getattr(never, "_tt_pairs_t", None) is None
    for w, row, t in zip(ws, rows, ts):
        assert engine.weight_planes(w, 2).data_ptr() == row.data_ptr()          # the cache answers, same buffer
        assert torch.equal(row, ops.split_pairs(w.detach()))
        if t is not None:
            assert engine.weight_pairs_t(w).data_ptr() == t.data_ptr()
            assert torch.equal(t, ops.split_pairs_dual(w.detach())[0])
    ops._bump_param_epoch()                                                      # a raw-pointer update (AdamW / EMA): trainable + non-static frozen alike
    assert engine.refresh_pair_operands(ws) == len(ws)


def test_ksplit_partial_exchange_soak():
    """The K-split of gemm_pairs8's left-over tiles exchanges fp32 partials between workgroups (write-through stores, a per-(tile, wave) counter,
    the last arriver sums in slice order and resets the counter): 60 back-to-back launches alternating between two split shapes and an
    unsplit one must reproduce their first results bit for bit (a stale partial, a counter left non-zero or an order that depends on the
    finisher would show), and agree with the unsplit kernel to fp32 rounding."""
    from timetuning_amd import hip_ops as ops

    lib = ops._lib.load()
    cases = []
    for i, (M, N, K) in enumerate([(25216, 384, 1536), (25216, 768, 3072), (25216, 1152, 384)]):
        x, w, b = rnd(f"soak.x{i}", M, K).cuda(), rnd(f"soak.w{i}", N, K, scale=0.05).cuda(), rnd(f"soak.b{i}", N, scale=0.1).cuda()
        r0 = rnd(f"soak.r{i}", M, N).cuda()
        cases.append((ops.split_pairs(x), ops.split_pairs(w), b, r0))
    def run(c):
        xp, wp, b, r0 = c
        r = r0.clone()
        return ops.linear_fwd_pairs(xp, wp, b, residual=r, out=r)["y"]
    first = [run(c) for c in cases]
    for it in range(20):
        for c, f in zip(cases, first):
            assert torch.equal(run(c), f), f"launch {it}: result differs from the first one"
    ops.set_tuning_knob("TT_Q8_KSPLIT", 0)
    try:
        for c, f in zip(cases[:2], first[:2]):
            ref = run(c)
            assert (ref - f).abs().max() <= 2e-6 * ref.abs().max()
    finally:
        ops.set_tuning_knob("TT_Q8_KSPLIT", 1)
    torch.cuda.synchronize()


@pytest.mark.parametrize("Fr,D,patch,route", [(3, 384, 16, 0), (128, 384, 16, 8), (64, 768, 16, 8), (5, 384, 8, 0)])
def test_patch_embed_on_pairs(Fr, D, patch, route):
    """tt_patch_embed_fwd_pairs (prepare_tokens of the "f16x3" mode, dino_vision_transformer.py:166-171,236-247) against the fp64 conv: at
    the f32 bound and not worse than the f32-MFMA patch embedding; with a frame map; the large cases run on the persistent pair kernel."""
    from timetuning_amd import _lib, hip_ops as ops

    Hh = Ww = 224 if patch == 16 else 64
    n = (Hh // patch) * (Ww // patch)
    K = 3 * patch * patch
    img, w, b = rnd(f"pep.img{Fr}.{Hh}", Fr + 1, 3, Hh, Ww), rnd(f"pep.w{D}.{K}", D, K, scale=0.05), rnd(f"pep.b{D}", D)
    cls, pos = rnd(f"pep.cls{D}", D), rnd(f"pep.pos{n}.{D}", n + 1, D)
    fmap = torch.arange(Fr, dtype=torch.int32).flip(0) + 1
    assert _lib.load().tt_linear_fwd_pairs_route(Fr * (n + 1), D, K, 0, 1, 1, 1, 0, 0) == route
    tok = ops.patch_embed_fwd_pairs(img.cuda(), ops.split_pairs(w.cuda()), b.cuda(), cls.cuda(), pos.cuda(), patch, fmap.cuda()).cpu()
    src = img[fmap.long()]
    conv = F.conv2d(src.double(), w.double().view(D, 3, patch, patch), b.double(), stride=patch)
    ref = torch.cat([cls.double().expand(Fr, 1, D), conv.flatten(2).transpose(1, 2)], 1) + pos.double()
    assert rel_err(tok, ref) < TOL_F32 and rel_l2(tok, ref) < TOL_L2
    assert (tok[:, 0] - (cls + pos[0])).abs().max().item() < 1e-6          # the class row: cls + pos[0] (- bias + bias in fp32)
    tok32 = ops.patch_embed_fwd(img.cuda(), w.cuda(), b.cuda(), cls.cuda(), pos.cuda(), patch, fmap.cuda()).cpu()
    assert rel_l2(tok, ref) <= rel_l2(tok32, ref)


# ---- round 5: the mode's RANGE contract.  The "f16x3" arithmetic shares fp32's precision, not its range: an operand
# beyond |x| = 65504 (or a non-finite one) has hi = inf / NaN.  Every kernel that produces pairs from fp32 values reports that through the
# caller's range flag; the Python host turns it into PairRangeError at the next ``check_pair_range`` (TimeT.train_update / the driver).
def _flag_clear(ops):
    f = ops.range_flag()
    f.zero_()
    return f


def test_pair_range_flag_every_producer():
    from timetuning_amd import hip_ops as ops

    f = _flag_clear(ops)
    # in range: the largest finite fp16, a value that ROUNDS to it, tiny and zero values - no flag
    x = rnd("rng.x", 64, 96, scale=3.0)
    x[0, :5] = torch.tensor([65504.0, -65519.0, 1e-30, 0.0, -6e-8])
    ops.split_pairs(x.cuda())
    ops.check_pair_range()
    assert int(f[0].item()) == 0
    # beyond it: 65520 is the first value that rounds to inf; inf and NaN themselves
    for bad in (65520.0, -1e6, float("inf"), float("nan")):
        y = x.clone(); y[7, 11] = bad
        ops.split_pairs(y.cuda())
        with pytest.raises(ops.PairRangeError, match="--precision f32"):
            ops.check_pair_range()
        assert int(f[0].item()) == 0          # (reset by the check)
    # LayerNorm writing pairs: a gamma of 1e5 pushes its output out of range (both kernels: vectorised D = 384, general D = 96)
    for D in (384, 96):
        xl = rnd(f"rng.ln{D}", 2, 17, D)
        ops.layernorm_fwd_pairs(xl.cuda(), torch.ones(D).cuda(), torch.zeros(D).cuda())
        ops.check_pair_range()
        g = torch.ones(D); g[3] = 1e5
        ops.layernorm_fwd_pairs(xl.cuda(), g.cuda(), torch.zeros(D).cuda())
        with pytest.raises(ops.PairRangeError):
            ops.check_pair_range()
    # a Linear whose PAIR output overflows although its operands are in range (the persistent kernel and the general one); its fp32 output
    # does not raise the flag (fp32 holds 1e6)
    for M, N, K in ((25216, 384, 384), (591, 256, 128)):
        xa, w = rnd(f"rng.lx{M}", M, K), rnd(f"rng.lw{N}", N, K, scale=0.05)
        xa[5, :] = 300.0; w[9, :] = 300.0     # y[5, 9] = 9e4 K ... far beyond 65504; both operands fine
        xp, wp = ops.split_pairs(xa.cuda()), ops.split_pairs(w.cuda())
        ops.check_pair_range()
        o = ops.linear_fwd_pairs(xp, wp, None)
        ops.check_pair_range()
        assert torch.isfinite(o["y"]).all() and o["y"][5, 9].item() > 65504
        ops.linear_fwd_pairs(xp, wp, None, out_f32=False, out_pairs=True)
        with pytest.raises(ops.PairRangeError):
            ops.check_pair_range()
    # gradients: a dy far BELOW fp16's range is scaled into it (no flag); a non-finite one is reported
    dy = rnd("rng.dy", 197, 128, scale=1e-7)
    ops.split_pairs_dual(dy.cuda(), want_row=True, want_colsum=True, scaled=True)
    ops.check_pair_range()
    dy[3, 3] = float("inf")
    ops.split_pairs_dual(dy.cuda(), want_row=True, want_colsum=True, scaled=True)
    with pytest.raises(ops.PairRangeError):
        ops.check_pair_range()
    # the batched weight refresh and the patch embedding
    wbig = rnd("rng.w", 128, 64); wbig[0, 0] = 1e5
    ops.split_pairs_dual_multi([(wbig.cuda(), torch.empty(128, 128, dtype=torch.float16, device="cuda"), None)])
    with pytest.raises(ops.PairRangeError):
        ops.check_pair_range()
    img = rnd("rng.img", 2, 3, 32, 48); img[1, 2, 5, 7] = 2e5
    wpe = ops.split_pairs(rnd("rng.pw", 128, 768, scale=0.02).cuda())
    ops.patch_embed_fwd_pairs(img.cuda(), wpe, torch.zeros(128).cuda(), torch.zeros(128).cuda(), torch.zeros(7, 128).cuda(), 16)
    with pytest.raises(ops.PairRangeError):
        ops.check_pair_range()
    ops.check_pair_range()   # clean again


def test_training_step_reports_out_of_range_weights():
    """End to end: one weight of a trainable block beyond fp16's range.  The f32 mode trains on; the "f16x3" mode's step is reported at the
    model surface (TimeT.check_pair_range, what train_update and the driver call) instead of silently turning the loss into NaN."""
    from timetuning_amd import hip_ops as ops
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.time_tuning import TimeT

    cfg = synth.ARCHS["tiny-s16"]
    fe = FeatureExtractor("dino-s16", "", [128, 128, 64, 32], unfreeze_layers=["blocks.11", "blocks.10"], vit_cfg=cfg, init="stress", return_attention=False)
    model = TimeT(fe, 20, prototype_init=torch.from_numpy(synth.make_prototypes(20, fe.feature_dim))).cuda()
    x = torch.from_numpy(synth.make_clips(2, 3, 224, seed=1)).cuda()
    with torch.no_grad():
        model.feature_extractor.backbone.blocks[3].mlp.fc1.weight[0, 0] = 1.0e5   # (a frozen block: its row pairs are made once)
    keep, ops.PAIRS_MIN_ROWS = ops.PAIRS_MIN_ROWS, 0
    try:
        _flag_clear(ops)
        loss = model(x, None, True, False)
        assert np.isfinite(loss.item())            # fp32 arithmetic: 1e5 is an ordinary number
        model.check_pair_range()                   # nothing to report in the f32 mode
        ops.set_gemm_precision("f16x3")
        model(x, None, True, False)
        with pytest.raises(ops.PairRangeError, match="--precision f32"):
            model.check_pair_range()
    finally:
        ops.set_gemm_precision("f32")
        ops.PAIRS_MIN_ROWS = keep
        _flag_clear(ops)


def _loguniform(name, *shape, lo=1e-7, hi=1e4):
    g = synth._philox("pairs." + name, 1)   # (the portable counter-based generator of every synthetic tensor)
    u, sgn = torch.from_numpy(g.random(size=tuple(shape))), torch.from_numpy(g.random(size=tuple(shape)))
    mag = torch.exp(np.log(lo) + u * (np.log(hi) - np.log(lo)))
    return (mag * torch.where(sgn < 0.5, -1.0, 1.0)).float()


@pytest.mark.parametrize("M,N,K", [(25216, 1152, 384), (25216, 384, 1536), (6304, 384, 384), (788, 384, 384)])
@pytest.mark.parametrize("kind", ["loguniform", "outlier_channels", "near_limit"])
def test_linear_pairs_heavy_tailed_operands(M, N, K, kind):
    """VERDICT r4 (weak 1a): `test_linear_pairs` feeds N(0, 1) x N(0, 0.05^2) only.  The same rule - error against fp64 not above the
    exact-f32 MFMA kernel's own on the same operands - on operands that stress the FORMAT: magnitudes log-uniform over eleven decades
    (1e-7 ... 1e4: from fp16-subnormal hi to near its top), activations with outlier channels (x 1e3, what DINO ViT residual streams
    carry), and operands at the edge of the range (|x| up to 6e4).  Nothing here may raise the range flag."""
    from timetuning_amd import hip_ops as ops

    w, b = rnd(f"ht.w{N}.{K}", N, K, scale=0.05), rnd(f"ht.b{N}", N, scale=0.1)
    if kind == "loguniform":
        x = _loguniform(f"ht.x{M}.{K}", M, K)
    elif kind == "outlier_channels":
        x = rnd(f"ht.x{M}.{K}", M, K)
        x[:, [3, 77, K - 5]] *= 1.0e3
    else:
        x = rnd(f"ht.x{M}.{K}", M, K)
        x = x * (6.0e4 / x.abs().max())
        w = w * (2.0e-2 / w.abs().max())   # keeps the PRODUCT finite in fp32 too; the operands' own magnitudes are what is tested
    _flag_clear(ops)
    xp, wp = ops.split_pairs(x.cuda()), ops.split_pairs(w.cuda())
    y = ops.linear_fwd_pairs(xp, wp, b.cuda())["y"].cpu()
    ops.check_pair_range()
    ref = x.double() @ w.double().t() + b.double()
    y32 = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda()).cpu()
    e_pair, l_pair, e32, l32 = rel_err(y, ref), rel_l2(y, ref), rel_err(y32, ref), rel_l2(y32, ref)
    import os
    if os.environ.get("TT_TEST_PRINT_ERRORS"):
        print(f"[errors] heavy-tailed {kind} {M}x{N}x{K}: f16x3 max-norm {e_pair:.2e} rel-L2 {l_pair:.2e} | f32 MFMA {e32:.2e} {l32:.2e}")
    assert e_pair < TOL_F32 and l_pair < TOL_L2
    assert l_pair <= l32 and e_pair <= 1.05 * e32


def test_linear_pairs_launch_neither_allocates_nor_breaks_capture():
    """ABI 7: the K-split workspace of the persistent kernels is the caller's - a launch allocates nothing (device
    memory unchanged across launches on a fresh stream, where the round-4 library hipMalloc'ed 32 MB per stream) and can be captured into a
    hipGraph whose replay leaves the eager launch's bits (fc2's shape takes the K-split: 41 left-over tiles x 6 slices)."""
    from timetuning_amd import hip_ops as ops

    M, N, K = 25216, 384, 1536
    xp, wp = ops.split_pairs(rnd("cap.x", M, K).cuda()), ops.split_pairs(rnd("cap.w", N, K, scale=0.05).cuda())
    b, r0 = rnd("cap.b", N).cuda(), rnd("cap.r", M, N).cuda()
    ref = r0.clone()
    ops.linear_fwd_pairs(xp, wp, b, residual=ref, out=ref)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.ksplit_workspace()                      # this stream's workspace: allocated and initialised OUTSIDE the launch path
        out = r0.clone()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(3):
            out.copy_(r0)
            ops.linear_fwd_pairs(xp, wp, b, residual=out, out=out)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, "a launch changed the device's free memory"
        assert torch.equal(out, ref)
        # capture + replay
        g = torch.cuda.CUDAGraph()
        buf = r0.clone()
        with torch.cuda.graph(g, stream=side):
            ops.linear_fwd_pairs(xp, wp, b, residual=buf, out=buf)
        for _ in range(3):
            buf.copy_(r0)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(buf, ref)
    # without a workspace the same call does not split K: same values to fp32 rounding, no error
    lib = __import__("timetuning_amd._lib", fromlist=["load"]).load()
    y2 = r0.clone()
    rc = lib.tt_linear_fwd_pairs(xp.data_ptr(), wp.data_ptr(), b.data_ptr(), y2.data_ptr(), y2.data_ptr(), None, None, M, N, K, 0, None, 0, None,
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_err(y2.cpu(), ref.cpu()) < 1e-6
