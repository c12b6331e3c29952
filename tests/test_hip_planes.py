"""Per-op parity of the bf16-plane kernels (gemm_planes.hip, attention_bf16.hip, layernorm_fwd_planes) through the C ABI.

planes = 1 is BASELINE C4's bf16 path: checked against an fp64 product of the SAME bf16-rounded operands (so only the fp32
accumulation differs) and, loosely, against the fp32 operands.  planes = 3 is the fp32-accurate split mode: held to the SAME
2e-5 bound as the f32-MFMA kernels (tests/test_hip_ops.py) against fp64 of the fp32 operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from timetuning_amd import synth

pytestmark = pytest.mark.gpu
TOL_F32 = 2e-5


def rnd(name, *shape, scale=1.0):
    return torch.from_numpy(synth.normal("planes." + name, shape, scale))


def planes_to_f64(p):
    return p.double().sum(0)


@pytest.mark.parametrize("planes", [1, 2, 3])
def test_split_planes(planes):
    from timetuning_amd import hip_ops as ops

    x = rnd("split", 64, 200, scale=3.0)
    x[0, :4] = torch.tensor([0.0, 1e-30, -65504.0, 3.0e38])
    p = ops.split_planes(x.cuda(), planes).cpu()
    assert p.shape == (planes, 64, 200) and p.dtype == torch.bfloat16
    assert torch.equal(p[0], x.to(torch.bfloat16))
    err = (planes_to_f64(p) - x.double()).abs() / x.double().abs().clamp_min(1e-20)
    bound = {1: 2.0 ** -8, 2: 2.0 ** -16, 3: 0.0}[planes]
    assert err[:, 4:].max().item() <= bound and err[0, :4].max().item() <= max(bound, 2.0 ** -16)   # (1e-30: bf16 subnormal range)


@pytest.mark.parametrize("planes", [1, 3])
def test_layernorm_planes(planes):
    from timetuning_amd import hip_ops as ops

    x, g, b = rnd("ln.x", 3, 197, 384, scale=2.0), 1.0 + 0.1 * rnd("ln.g", 384), 0.1 * rnd("ln.b", 384)
    ref = F.layer_norm(x.double(), (384,), g.double(), b.double(), 1e-6)
    y, mean, rstd = ops.layernorm_fwd_planes(x.cuda(), g.cuda(), b.cuda(), planes, save_stats=True)
    assert y.shape == (planes, 591, 384)
    assert rel_err(planes_to_f64(y.cpu()).view(3, 197, 384), ref) < (TOL_F32 if planes == 3 else 5e-3)
    assert rel_err(mean.cpu(), x.double().mean(-1).view(-1)) < 1e-5
    yd = ops.layernorm_fwd_planes(x.cuda(), g.cuda(), b.cuda(), planes, drop_first_token=True)
    assert yd.shape == (planes, 3 * 196, 384)
    assert rel_err(planes_to_f64(yd.cpu()).view(3, 196, 384), ref[:, 1:]) < (TOL_F32 if planes == 3 else 5e-3)


@pytest.mark.parametrize("planes", [1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(788, 384, 384), (1000, 1152, 384), (8192, 1536, 384), (394, 64, 1536), (2048, 768, 3072)])
def test_linear_planes(planes, M, N, K):
    from timetuning_amd import hip_ops as ops

    x, w, b = rnd(f"lin.x{M}", M, K), rnd(f"lin.w{N}.{K}", N, K, scale=0.05), rnd(f"lin.b{N}", N, scale=0.1)
    xp, wp = ops.split_planes(x.cuda(), planes), ops.split_planes(w.cuda(), planes)
    o = ops.linear_fwd_planes(xp, wp, b.cuda())
    ref32 = x.double() @ w.double().t() + b.double()
    if planes == 3:
        assert rel_err(o["y"].cpu(), ref32) < TOL_F32
    elif planes == 2:
        assert rel_err(o["y"].cpu(), ref32) < 2e-4
    else:
        ref_bf = planes_to_f64(xp.cpu()) @ planes_to_f64(wp.cpu()).t() + b.double()   # the same rounded operands, exact products
        assert rel_err(o["y"].cpu(), ref_bf) < TOL_F32
        assert rel_err(o["y"].cpu(), ref32) < 2e-2


@pytest.mark.parametrize("planes", [1, 3])
def test_linear_planes_epilogues(planes):
    """bias + GELU + pre-activation + plane output; residual add in place; plane-only output."""
    from timetuning_amd import hip_ops as ops

    M, N, K = 591, 256, 128
    x, w, b, res = rnd("epi.x", M, K), rnd("epi.w", N, K, scale=0.1), rnd("epi.b", N, scale=0.1), rnd("epi.r", M, N)
    xp, wp = ops.split_planes(x.cuda(), planes), ops.split_planes(w.cuda(), planes)
    pre_ref = planes_to_f64(xp.cpu()) @ planes_to_f64(wp.cpu()).t() + b.double()
    tol = TOL_F32 if planes == 3 else 2e-5
    o = ops.linear_fwd_planes(xp, wp, b.cuda(), act=1, out_planes=planes, save_pre=True)
    assert rel_err(o["pre"].cpu(), pre_ref) < tol
    assert rel_err(o["y"].cpu(), F.gelu(pre_ref)) < tol
    got = planes_to_f64(o["planes"].cpu())
    assert rel_err(got, o["y"].cpu().double()) <= (0.0 if planes == 3 else 2.0 ** -8)   # the planes ARE the fp32 result, split
    rc = res.clone().cuda()
    o2 = ops.linear_fwd_planes(xp, wp, b.cuda(), residual=rc, out=rc)
    assert o2["y"].data_ptr() == rc.data_ptr() and rel_err(rc.cpu(), pre_ref + res.double()) < tol
    o3 = ops.linear_fwd_planes(xp, wp, None, out_f32=False, out_planes=planes)
    assert o3["y"] is None and rel_err(planes_to_f64(o3["planes"].cpu()), pre_ref - b.double()) < (tol if planes == 3 else 5e-3)


@pytest.mark.parametrize("Fr,N,H", [(3, 197, 6), (2, 50, 2), (1, 256, 12), (2, 225, 3), (5, 17, 1)])
def test_attention_bf16(Fr, N, H):
    from timetuning_amd import hip_ops as ops

    D = 64 * H
    qkv = (rnd(f"att.{Fr}.{N}.{H}", Fr, N, 3 * D) * 0.7).to(torch.bfloat16)
    out = ops.attention_fwd_bf16(qkv.cuda(), H).cpu()
    q, k, v = qkv.double().view(Fr, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) * 64 ** -0.5, dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(Fr, N, D)
    # P and the output are rounded to bf16 (8 significant bits each)
    assert rel_err(out.double(), ref) < 1.5e-2
    assert (out.double() - ref).abs().mean().item() < 2e-3 * ref.abs().mean().item() + 1e-4


@pytest.mark.parametrize("M,N,K", [(3152, 768, 3072), (6304, 1536, 384), (591, 256, 512), (6304, 384, 1152)])
def test_backward_products_on_bf16_planes(M, N, K):
    """dx = dy @ w (* gelu'(pre)) and dw = dy^T @ x on bf16 operands (the "bf16" mode's backward, BASELINE C4): against fp64
    products of the SAME bf16-rounded operands at the f32 bound (only the fp32 accumulation differs; split-K fold included),
    and loosely against the fp32 operands."""
    from timetuning_amd import hip_ops as ops

    dy, w, x = rnd(f"bwd.dy{M}.{N}", M, N, scale=0.1), rnd(f"bwd.w{N}.{K}", N, K, scale=0.05), rnd(f"bwd.x{M}.{K}", M, K)
    pre = rnd(f"bwd.pre{M}.{K}", M, K)
    dyb, wb, xb = dy.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), x.to(torch.bfloat16).double()
    assert ops.bwd_planes_ok(M, N, K)
    dx = ops.linear_bwd_data_planes(dy.cuda(), w.cuda())
    assert rel_err(dx.cpu(), dyb @ wb) < TOL_F32
    assert rel_err(dx.cpu(), dy.double() @ w.double()) < 2e-2
    pd = pre.double().requires_grad_(True)
    F.gelu(pd).sum().backward()
    dxg = ops.linear_bwd_data_planes(dy.cuda(), w.cuda(), pre.cuda())
    assert rel_err(dxg.cpu(), (dyb @ wb) * pd.grad) < TOL_F32
    dw, db = ops.linear_bwd_weight_planes(dy.cuda(), x.cuda())
    assert rel_err(dw.cpu(), dyb.t() @ xb) < TOL_F32
    assert rel_err(dw.cpu(), dy.double().t() @ x.double()) < 2e-2
    assert rel_err(db.cpu(), dy.double().sum(0)) < 1e-5
    t = ops.transpose_planes(x.cuda())
    assert t.shape == (1, K, (M + 63) // 64 * 64) and torch.equal(t[0, :, :M].cpu(), x.to(torch.bfloat16).t())
    assert (t[0, :, M:] == 0).all()


# ---- gemm_planes8_kernel (round 3): the persistent 8-phase kernel behind tt_linear_fwd_planes ------------------------------------------

P8_SHAPES = {   # tile counts (256-row x 256 / 128-column tiles): what the work decomposition does with them on 256 CUs
    "one round + 82 half tiles": (25216, 3, 256),
    "ragged M, round-robin deal of 202 tiles": (197 * 130 + 7, 2, 384),
    "half tiles only (128 tiles)": (32768, 1, 256),
    "two whole rounds": (16384, 8, 128),
    "three rounds + 246 half tiles": (25216, 9, 256),
}


@pytest.mark.parametrize("planes", [1, 3])
@pytest.mark.parametrize("case", list(P8_SHAPES))
def test_linear_planes8_every_epilogue(planes, case, monkeypatch):
    """The persistent kernel on every tile-count regime of its work decomposition x every compiled epilogue (fp32, fp32 + residual in
    place, bf16 / 3-plane output with and without GELU): (1) the route query says it runs, (2) fp64 products of the same plane operands
    on a row sample that includes the first rows, the LAST rows (ragged tail, half tiles) and random ones, (3) the WHOLE output against
    gemm_planes_kernel (TT_PLANES_VARIANT=10) - identical bits at one plane without GELU, (4) a second run equals the first bit for bit
    (the counted-vmcnt schedule has no run-to-run freedom)."""
    from timetuning_amd import _lib, hip_ops as ops

    M, strips, K = P8_SHAPES[case]
    N = strips * (256 if planes == 1 else 128)
    g = torch.Generator().manual_seed(100 * planes + list(P8_SHAPES).index(case))
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    xp, wp = ops.split_planes(x, planes), ops.split_planes(w, planes)
    idx = torch.cat([torch.arange(0, 200), torch.arange(M - 200, M), torch.randint(0, M, (200,), generator=g)]).cuda()
    ref = planes_to_f64(xp[:, idx]) @ planes_to_f64(wp).t() + b.double()
    lib = _lib.load()
    epilogues = [dict(act=0, po=0, res=False), dict(act=0, po=0, res=True)]
    epilogues += [dict(act=0, po=1, res=False), dict(act=1, po=1, res=False)] if planes == 1 else [dict(act=1, po=3, res=False)]
    for e in epilogues:
        r = torch.randn(M, N, generator=g).cuda() if e["res"] else None

        def run():
            rr = r.clone() if r is not None else None
            o = ops.linear_fwd_planes(xp, wp, b, residual=rr, act=e["act"], out_f32=e["po"] == 0, out_planes=e["po"], out=rr)
            return o["y"] if e["po"] == 0 else o["planes"]

        ops.set_tuning_knob("TT_PLANES_VARIANT", 0)
        assert lib.tt_linear_fwd_planes_route(planes, M, N, K, e["act"], 1, int(e["res"]), int(e["po"] == 0), e["po"], 0) == 8, (case, e)
        new = run()
        again = run()
        assert torch.equal(new, again), (case, e)
        try:
            ops.set_tuning_knob("TT_PLANES_VARIANT", 10)   # (the library reads its knobs once; tests flip them through the setter)
            assert lib.tt_linear_fwd_planes_route(planes, M, N, K, e["act"], 1, int(e["res"]), int(e["po"] == 0), e["po"], 0) == 0
            old = run()
        finally:
            ops.set_tuning_knob("TT_PLANES_VARIANT", 0)
        want = torch.nn.functional.gelu(ref) if e["act"] else ref
        if e["res"]:
            want = want + r.double()[idx]
        got = new.double()[idx] if e["po"] == 0 else new.double().sum(0)[idx]
        tol = TOL_F32 if (planes == 3 or e["po"] == 0) else 5e-3          # bf16 output: 2^-9 rounding (and the tanh-form GELU, 5e-4 absolute)
        assert rel_err(got.cpu(), want.cpu()) < tol, (case, e)
        a, c = (new, old) if e["po"] == 0 else (new.float().sum(0), old.float().sum(0))
        if planes == 1 and not e["act"]:
            assert torch.equal(a, c), (case, e)                            # same products in the same order: identical bits
        else:
            lim = 4e-3 if planes == 1 else 2e-6                            # P = 1 GELU: the SAME tanh form on both routes (round 4); at most a flipped bf16 rounding (2^-8)
            assert ((a - c).abs().max() / c.abs().max()).item() < lim, (case, e)


def test_linear_planes_route_and_fallbacks():
    """What does NOT go to the persistent kernel keeps working on gemm_planes_kernel: a pre-activation output, fp32 y together with plane
    outputs, too few tiles, a column count that is not a whole tile, no bias."""
    from timetuning_amd import _lib, hip_ops as ops

    lib = _lib.load()
    route = lambda P, M, N, K, act=0, bias=1, res=0, y=1, po=0, pre=0: lib.tt_linear_fwd_planes_route(P, M, N, K, act, bias, res, y, po, pre)
    assert route(1, 25216, 2304, 768, y=0, po=1) == 8 and route(3, 25216, 1152, 384) == 8
    assert route(1, 25216, 2304, 768, y=0, po=1, pre=1) == 0          # pre-activation wanted
    assert route(1, 25216, 2304, 768, y=1, po=1) == 0                 # y and planes together
    assert route(1, 25216, 2304, 768, bias=0) == 0
    assert route(1, 2048, 768, 3072) == 0                             # 24 tiles
    assert route(1, 25216, 1152, 768) == 0 and route(3, 25216, 1152 + 64, 384) == 0   # not whole column tiles
    assert route(2, 25216, 2304, 768) == 0                            # two-plane mode
    x, w, b = rnd("rt.x", 2048, 128), rnd("rt.w", 768, 128, scale=0.05), rnd("rt.b", 768, scale=0.1)
    o = ops.linear_fwd_planes(ops.split_planes(x.cuda(), 3), ops.split_planes(w.cuda(), 3), b.cuda(), act=1, save_pre=True, out_planes=3)
    ref = x.double() @ w.double().t() + b.double()
    assert rel_err(o["pre"].cpu(), ref) < TOL_F32 and rel_err(planes_to_f64(o["planes"].cpu()), F.gelu(ref)) < TOL_F32


@pytest.mark.parametrize("Fr,D,route", [(3, 768, 0), (64, 768, 8), (5, 384, 0)])
def test_patch_embed_on_bf16_operands(Fr, D, route):
    """tt_patch_embed_fwd_planes (BASELINE C4's bf16 path of dino_vision_transformer.py:166-171,236-247) against the fp64 conv of the
    SAME bf16-rounded operands (only the fp32 accumulation differs), loosely against the fp32 operands, with a frame map; the large
    case runs on the persistent 8-phase kernel, the small ones on gemm_planes_kernel."""
    from timetuning_amd import _lib, hip_ops as ops

    patch, Hh, Ww = 16, 224, 224
    n = (Hh // patch) * (Ww // patch)
    img, w, b = rnd("pe.img", Fr + 1, 3, Hh, Ww), rnd("pe.w", D, 3 * patch * patch, scale=0.05), rnd("pe.b", D)
    cls, pos = rnd("pe.cls", D), rnd("pe.pos", n + 1, D)
    fmap = torch.arange(Fr, dtype=torch.int32).flip(0) + 1
    assert _lib.load().tt_linear_fwd_planes_route(1, Fr * (n + 1), D, 768, 0, 1, 1, 1, 0, 0) == route
    wp = ops.split_planes(w.cuda(), 1)
    tok = ops.patch_embed_fwd_planes(img.cuda(), wp, b.cuda(), cls.cuda(), pos.cuda(), patch, fmap.cuda()).cpu()
    src = img[fmap.long()]
    conv = F.conv2d(src.to(torch.bfloat16).double(), w.to(torch.bfloat16).double().view(D, 3, patch, patch), b.double(), stride=patch)
    ref = torch.cat([cls.double().expand(Fr, 1, D), conv.flatten(2).transpose(1, 2)], 1) + pos.double()
    assert rel_err(tok, ref) < TOL_F32
    assert (tok[:, 0] - (cls + pos[0])).abs().max().item() < 1e-6          # the class row: cls + pos[0] (- bias + bias in fp32)
    full = F.conv2d(src.double(), w.double().view(D, 3, patch, patch), b.double(), stride=patch).flatten(2).transpose(1, 2) + pos.double()[1:]
    assert rel_err(tok[:, 1:], full) < 1e-2
    tok32 = ops.patch_embed_fwd(img.cuda(), w.cuda(), b.cuda(), cls.cuda(), pos.cuda(), patch, fmap.cuda()).cpu()
    assert rel_err(tok, tok32.double()) < 1e-2


@pytest.mark.parametrize("planes", [1, 3])
def test_planes8_load_part_orders_give_the_same_bits(planes, monkeypatch):
    """TT_P8_ORDER (a tuning knob: ops.set_tuning_knob) only moves a wave's DMA instructions relative to its fragment reads inside a phase (gemm_planes8.hip `reads_first`):
    every order must leave the same bits; the default (3) is checked against fp64 by test_linear_planes8_every_epilogue."""
    from timetuning_amd import hip_ops as ops

    BN = 256 if planes == 1 else 128
    M, N, K = 25216, 3 * BN, 512
    xp, wp = ops.split_planes(rnd("ord.x", M, K).cuda(), planes), ops.split_planes(rnd("ord.w", N, K, scale=0.05).cuda(), planes)
    b = rnd("ord.b", N).cuda()
    outs = []
    try:
        for order in (3, 0, 1, 2):
            ops.set_tuning_knob("TT_P8_ORDER", order)
            outs.append(ops.linear_fwd_planes(xp, wp, b, act=1, out_f32=False, out_planes=planes)["planes"].clone())
    finally:
        ops.set_tuning_knob("TT_P8_ORDER", 3)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])


@pytest.mark.parametrize("Fr,N,H", [(2, 197, 12), (3, 50, 2), (1, 256, 3), (2, 17, 1)])
def test_attention_backward_on_bf16_products(Fr, N, H):
    """tt_attention_bwd_bf16 (the bf16 path's attention backward: products on v_mfma_f32_16x16x16_bf16, softmax statistics / P / dS in fp32)
    against fp64 autograd of the fp32 inputs at a bf16 bound, and against the exact-f32 kernel (close, not equal)."""
    from timetuning_amd import hip_ops as ops

    qkv = rnd("ab.qkv", Fr, N, 3 * H * 64, scale=0.6).cuda()
    dout = rnd("ab.do", Fr, N, H * 64).cuda()
    out, lse, _ = ops.attention_fwd(qkv, H, save_lse=True)
    d32 = ops.attention_bwd(qkv, out, dout, lse, H)
    d16 = ops.attention_bwd(qkv, out, dout, lse, H, bf16_products=True)
    qt = qkv.double().cpu().requires_grad_(True)
    q, k, v = qt.view(Fr, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(Fr, N, H * 64)
    (o * dout.double().cpu()).sum().backward()
    assert rel_err(d32, qt.grad) < TOL_F32
    e16 = rel_err(d16, qt.grad)
    assert 1e-5 < e16 < 2e-2, e16
    cos = torch.nn.functional.cosine_similarity(d16.double().cpu().flatten(), qt.grad.flatten(), dim=0).item()
    assert cos > 0.9999, cos


def test_label_propagation_similarities_on_bf16_products():
    """In the "bf16" mode (``precision`` = TT_PRECISION_BF16, BASELINE C4's path) the label propagation's cosine similarities run on bf16 MFMA
    (gemm_f32_kernel<BF16>): the propagated maps must equal the oracle's on the bf16-ROUNDED features (mask_propagation.py:448-496) as
    well as the f32 mode's equal the oracle's on the fp32 features, and the mode must actually change the similarities."""
    import torch.nn.functional as F_
    from oracle import timet_oracle as O
    from timetuning_amd import hip_ops as ops

    fs, bs, g, D, K = 4, 2, 14, 768, 21
    n = g * g
    feats = rnd("lp.f", fs, bs, n, D)
    for t in range(1, fs):
        feats[t] = 0.7 * feats[t - 1] + 0.3 * feats[t]
    xn = F_.normalize(feats, dim=-1)
    seg0 = torch.softmax(rnd("lp.s", bs, n, K) * 2, -1)

    def off_fraction(maps, x_ref):
        bad = tot = 0
        for b in range(bs):
            seed = seg0[b].view(g, g, K).permute(2, 0, 1).unsqueeze(0)
            ref = torch.stack(O.propagate_labels(3, 6, 5, g, x_ref[:, b], seed)).reshape(fs - 1, K, n).transpose(1, 2).numpy()
            d = np.abs(maps[:, b] - ref).max(-1) > 1e-5 * np.abs(ref).max()
            bad += int(d.sum()); tot += d.size
        return bad / tot

    m32 = ops.label_propagate_maps(xn.cuda(), seg0.cuda(), 3, 6, 5, 0.1).cpu().numpy()
    try:
        ops.set_gemm_precision("bf16")
        m16 = ops.label_propagate_maps(xn.cuda(), seg0.cuda(), 3, 6, 5, 0.1).cpu().numpy()
    finally:
        ops.set_gemm_precision("f32")
    assert off_fraction(m32, xn) < 0.01
    assert off_fraction(m16, xn.to(torch.bfloat16).float()) < 0.01
    assert not np.array_equal(m16, m32)


@pytest.mark.parametrize("R,C", [(6304, 768), (591, 256), (100, 64), (3152, 2304)])
def test_transpose_planes_with_column_sums(R, C):
    """tt_transpose_planes_colsum: the transposed bf16 image is the plain entry point's bit for bit, the fp32 column sums (the bias
    gradient of the bf16 path's weight-gradient product) match fp64 at fp32 accuracy and are the same on every run."""
    from timetuning_amd import hip_ops as ops

    x = rnd("tc.x", R, C).cuda()
    plain = ops.transpose_planes(x)
    t, s = ops.transpose_planes(x, colsum=True)
    assert torch.equal(t, plain)
    assert rel_err(s, x.double().sum(0)) < 2e-6
    assert torch.equal(ops.transpose_planes(x, colsum=True)[1], s)
