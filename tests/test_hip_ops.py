"""GPU parity tests, one per C-ABI op, against the CPU oracle / plain torch fp64 on seeded inputs.
Every call goes through libtimetuning_hip.so (ctypes); nothing here can pass on a CPU fallback.
Tolerance: the north-star bound is 1e-3 relative fp32; the kernels are exact-f32 MFMA / VALU, so the
tests hold them to 2e-5 (measured headroom) unless stated."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, rel_l2
from oracle import timet_oracle as O
from timetuning_amd import synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from timetuning_amd import hip_ops

    return hip_ops


def dev(a):
    return torch.as_tensor(a).cuda().contiguous()


def rnd(name, *shape, std=1.0):
    return torch.from_numpy(synth.normal("t." + name, shape, std))


@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (591, 1152, 384), (130, 72, 20), (64, 64, 64), (257, 200, 256), (25216, 384, 1536),
                                   (392, 50, 256), (77, 21, 30), (5, 3, 2), (6304, 1536, 384), (3, 128, 64), (18912, 384, 384),
                                   (788, 1536, 384), (788, 384, 1536), (788, 1152, 384), (394, 1024, 1024)])   # small grids: BK = 64 slabs
def test_linear_fwd(ops, M, N, K):
    x, w, b, r = rnd("lx", M, K), rnd("lw", N, K, std=0.05), rnd("lb", N), rnd("lr", M, N)
    ref = F.linear(x.double(), w.double(), b.double())
    y = ops.linear_fwd(dev(x), dev(w), dev(b))
    assert rel_err(y.cpu(), ref) < TOL
    y2, pre = ops.linear_fwd(dev(x), dev(w), dev(b), residual=dev(r), act=1, save_pre=True)
    assert rel_err(pre.cpu(), ref) < TOL
    assert rel_err(y2.cpu(), F.gelu(ref) + r.double()) < TOL
    y3 = ops.linear_fwd(dev(x), dev(w))
    assert rel_err(y3.cpu(), F.linear(x.double(), w.double())) < TOL


@pytest.mark.parametrize("M,N,K", [(788, 384, 1536), (394, 1024, 384), (130, 72, 20), (6272, 200, 256), (392, 50, 256), (101, 21, 30),
                                   (6304, 1536, 384)])
def test_linear_bwd(ops, M, N, K):
    dy, w, x, pre = rnd("bdy", M, N), rnd("bw", N, K, std=0.05), rnd("bx", M, K), rnd("bpre", M, K)
    dx = ops.linear_bwd_data(dev(dy), dev(w))
    assert rel_err(dx.cpu(), dy.double() @ w.double()) < TOL
    pr = pre.double().requires_grad_(True)
    F.gelu(pr).backward(dy.double() @ w.double())
    dx2 = ops.linear_bwd_data(dev(dy), dev(w), gelu_pre=dev(pre))
    assert rel_err(dx2.cpu(), pr.grad) < TOL
    dw, db = ops.linear_bwd_weight(dev(dy), dev(x))
    assert rel_err(dw.cpu(), dy.double().T @ x.double()) < TOL
    assert rel_err(db.cpu(), dy.double().sum(0)) < TOL


def test_gemm_batched_and_layouts(ops):
    A, B = rnd("ga", 5, 196, 64), rnd("gb", 5, 196, 64)
    C = ops.gemm(dev(A), dev(B))
    assert rel_err(C.cpu(), A.double() @ B.double().transpose(1, 2)) < TOL
    A2, B2 = rnd("ga2", 300, 132), rnd("gb2", 300, 88)  # both stored [K][*]
    C2 = ops.gemm(dev(A2), dev(B2), a_mmajor=True, b_nmajor=True, alpha=0.5)
    assert rel_err(C2.cpu(), 0.5 * A2.double().T @ B2.double()) < TOL


@pytest.mark.parametrize("patch,D,Fr", [(16, 384, 5), (8, 64, 2)])
def test_patch_embed(ops, patch, D, Fr):
    n = (224 // patch) ** 2
    p = {"patch_embed.proj.weight": rnd("pw", D, 3, patch, patch, std=0.05), "patch_embed.proj.bias": rnd("pb", D),
         "cls_token": rnd("pc", 1, 1, D), "pos_embed": rnd("pp", 1, n + 1, D)}
    x = rnd("pimg", Fr, 3, 224, 224)
    ref = O.prepare_tokens({k: v.double() for k, v in p.items()}, x.double(), patch)
    tok = ops.patch_embed_fwd(dev(x), dev(p["patch_embed.proj.weight"].reshape(D, -1)), dev(p["patch_embed.proj.bias"]),
                              dev(p["cls_token"].reshape(D)), dev(p["pos_embed"].reshape(n + 1, D)), patch)
    assert rel_err(tok.cpu(), ref) < TOL
    fmap = torch.tensor([Fr - 1, 0, 1], dtype=torch.int32)
    tok2 = ops.patch_embed_fwd(dev(x), dev(p["patch_embed.proj.weight"].reshape(D, -1)), dev(p["patch_embed.proj.bias"]),
                               dev(p["cls_token"].reshape(D)), dev(p["pos_embed"].reshape(n + 1, D)), patch, frame_map=dev(fmap))
    assert rel_err(tok2.cpu(), ref[fmap.long()]) < TOL


@pytest.mark.parametrize("rows,D", [(788, 384), (37, 768), (5, 64), (3, 1000)])
def test_layernorm(ops, rows, D):
    x, g, b, dy = rnd("nx", rows, D) * 3 + 0.7, rnd("ng", D) * 0.1 + 1, rnd("nb", D), rnd("ndy", rows, D)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(xd, (D,), gd, bd, 1e-6)
    ref.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(g), dev(b), save_stats=True)
    assert rel_err(y.cpu(), ref.detach()) < TOL
    dx, dg, db = ops.layernorm_bwd(dev(dy), dev(x), dev(g), mean, rstd)
    assert rel_err(dx.cpu(), xd.grad) < 5e-5
    assert rel_err(dg.cpu(), gd.grad) < 5e-5
    assert rel_err(db.cpu(), bd.grad) < 5e-5
    acc = dev(rnd("nacc", rows, D))
    dx2, _, _ = ops.layernorm_bwd(dev(dy), dev(x), dev(g), mean, rstd, need_wgrad=False, dx_accum=acc.clone())
    assert rel_err(dx2.cpu(), xd.grad + acc.cpu().double()) < 5e-5


def _attn_ref(qkv, H):
    Fr, N, D3 = qkv.shape
    D = D3 // 3
    hd = D // H
    t = qkv.reshape(Fr, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    a = ((q @ k.transpose(-2, -1)) * hd ** -0.5)
    lse = torch.logsumexp(a, -1)
    p = a.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(Fr, N, D), lse, p


@pytest.mark.parametrize("Fr,N,H", [(3, 197, 6), (2, 50, 2), (1, 256, 1), (2, 120, 3), (1, 17, 12), (2, 785, 6), (1, 300, 2)])
def test_attention_fwd_bwd(ops, Fr, N, H):
    """N <= 256: register-resident score rows; N > 256 (ViT-S/8: 785 tokens): the KV-tiled online-softmax kernel."""
    qkv = rnd("aq", Fr, N, 3 * H * 64) * 1.5
    do = rnd("ado", Fr, N, H * 64)
    qd = qkv.double().requires_grad_(True)
    ref, lse_ref, p_ref = _attn_ref(qd, H)
    ref.backward(do.double())
    out, lse, probs = ops.attention_fwd(dev(qkv), H, save_lse=True, return_probs=True)
    assert rel_err(out.cpu(), ref.detach()) < TOL
    assert rel_err(lse.cpu(), lse_ref.detach()) < TOL
    assert rel_err(probs.cpu(), p_ref.detach()) < TOL   # N > 256: the row-per-wave probabilities kernel
    dqkv = ops.attention_bwd(dev(qkv), out, dev(do), lse, H)
    assert rel_err(dqkv.cpu(), qd.grad) < 5e-5
    # the "f16x3" mode's backward (tt_attention_bwd_pairs: S and dP on three fp16 MFMAs per term): the same bound, not worse than the
    # fp32-MFMA kernels by more than rounding noise - also on a gradient of 1e-7 (the power-of-two scale of dout) - and the same amax slot
    for gs in (1.0, 1e-7):
        e32 = rel_l2(ops.attention_bwd(dev(qkv), out, dev(do * gs), lse, H).cpu(), qd.grad * gs)
        slot = torch.zeros(ops.AmaxPool.get(torch.device("cuda")).SLOT, device="cuda")
        dq_p = ops.attention_bwd(dev(qkv), out, dev(do * gs), lse, H, amax_out=slot, pair_products=True)
        assert rel_err(dq_p.cpu(), qd.grad * gs) < 5e-5, gs
        assert rel_l2(dq_p.cpu(), qd.grad * gs) < max(2.0 * e32, 2e-6), (gs, rel_l2(dq_p.cpu(), qd.grad * gs), e32)
        assert slot.max().item() == dq_p.abs().max().item()
    ops.check_pair_range()


def test_attention_bwd_pairs_scale_sources_and_range(ops):
    """tt_attention_bwd_pairs: (1) max |dout| from a producer's amax slot gives the bits of the call's own max pass (the same power of
    two); (2) values beyond what the fp16 halves can carry - here |v| beyond 65504 - raise the pair range flag (``PairRangeError``)
    instead of passing silently."""
    from timetuning_amd import hip_ops
    Fr, N, H = 2, 197, 3
    qkv = dev(rnd("apq", Fr, N, 3 * H * 64) * 1.2)
    do = dev(rnd("apd", Fr, N, H * 64) * 3e-5)
    out, lse, _ = ops.attention_fwd(qkv, H, save_lse=True)
    own = ops.attention_bwd(qkv, out, do, lse, H, pair_products=True)
    slot = torch.zeros(ops.AmaxPool.get(torch.device("cuda")).SLOT, device="cuda")
    slot[::64][:16] = do.abs().max() * torch.tensor([1.0] + [0.3] * 15, device="cuda")   # the producer's 16 ways: their maximum is max |dout|
    from_slot = ops.attention_bwd(qkv, out, do, lse, H, pair_products=True, dout_amax=slot)
    assert torch.equal(own, from_slot)
    ops.check_pair_range()
    big = qkv.clone()
    big[..., 2 * H * 64:] *= 1e5            # v
    out_b, lse_b, _ = ops.attention_fwd(big, H, save_lse=True)
    ops.attention_bwd(big, out_b, do, lse_b, H, pair_products=True)
    with pytest.raises(hip_ops.PairRangeError):
        ops.check_pair_range()
    ops.check_pair_range()                  # (reset by the raise)


def test_l2norm(ops):
    x, dxn = rnd("l2x", 700, 256) * 4, rnd("l2d", 700, 256)
    xd = x.double().requires_grad_(True)
    ref = F.normalize(xd, dim=-1)
    ref.backward(dxn.double())
    xn, inv = ops.l2norm_fwd(dev(x), save_inv=True)
    assert rel_err(xn.cpu(), ref.detach()) < TOL
    dx = ops.l2norm_bwd(dev(dxn), xn, inv)
    assert rel_err(dx.cpu(), xd.grad) < TOL
    big = dev(rnd("l2s", 20, 3, 96))  # row-strided view: rows of the middle slice
    view = big[:, 1, :]
    assert rel_err(ops.l2norm_fwd(view).cpu(), F.normalize(view.cpu().double(), dim=-1)) < TOL
    w = dev(rnd("l2w", 200, 256))
    ref_w = F.normalize(w.cpu().double(), dim=1)
    ops.normalize_rows_(w)
    assert rel_err(w.cpu(), ref_w) < TOL


@pytest.mark.parametrize("M,N,K,gelu", [(6304, 384, 1536, True), (6304, 1536, 384, False), (6304, 1152, 384, False), (6304, 384, 384, False),
                                         (788, 256, 512, True), (400, 48, 64, False)])
def test_linear_bwd_fused_equals_the_two_products(ops, M, N, K, gelu):
    """tt_linear_bwd: both backward products of an nn.Linear in one call - one LAUNCH for the two GEMMs on the block / head shapes of the
    32 target frames (dgrad tiles + split-K weight-gradient slices in one grid) - must equal tt_linear_bwd_data + tt_linear_bwd_weight bit
    for bit and the fp64 products within the f32 bound."""
    dy, w, x = rnd(f"lbf.dy{M}.{N}", M, N), rnd(f"lbf.w{N}.{K}", N, K, std=0.05), rnd(f"lbf.x{M}.{K}", M, K)
    pre = rnd(f"lbf.pre{M}.{K}", M, K) if gelu else None
    dx, dw, db = ops.linear_bwd(dev(dy), dev(w), dev(x), gelu_pre=dev(pre) if gelu else None)
    dx2 = ops.linear_bwd_data(dev(dy), dev(w), gelu_pre=dev(pre) if gelu else None)
    dw2, db2 = ops.linear_bwd_weight(dev(dy), dev(x))
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    ref_dx = dy.double() @ w.double()
    if gelu:
        p64 = pre.double()
        ref_dx = ref_dx * (0.5 * (1 + torch.erf(p64 / 2 ** 0.5)) + p64 * torch.exp(-0.5 * p64 * p64) / (2 * np.pi) ** 0.5)
    assert rel_err(dx.cpu(), ref_dx) < 2e-5
    assert rel_err(dw.cpu(), dy.double().t() @ x.double()) < 2e-5
    assert rel_err(db.cpu(), dy.double().sum(0)) < 2e-5


def test_my_utils_sinkhorn_signature(golden):
    """``my_utils.sinkhorn(Q, nmb_iters, world_size)`` as the reference calls it (Q = exp(scores / eps).T, my_utils.py:246-274)."""
    from timetuning_amd.my_utils import sinkhorn

    g = golden("sinkhorn")
    assert rel_err(sinkhorn(dev(g["kat_in"]), 3).cpu(), g["kat_it3"]) < 1e-5
    assert rel_err(sinkhorn(dev(g["kat_in"]), 0).cpu(), g["kat_it0"]) < 1e-5
    for tag in "ac":
        Q = torch.exp(torch.from_numpy(g[f"{tag}_scores"]) / 0.05).t().contiguous()
        assert rel_err(sinkhorn(Q.cuda(), int(g[f"{tag}_iters"])).cpu(), g[f"{tag}_q"]) < 1e-4, tag
    # the positive matrix in [B, K] layout (what the all-gather of the ranks' columns yields), rank 1's rows of the W = 2 problem
    from timetuning_amd import hip_ops

    g2 = golden("sinkhorn_w2")
    E = torch.exp(torch.from_numpy(g2["scores"]) / 0.05).contiguous().cuda()
    q = hip_ops.sinkhorn_from_q(E, int(g2["iters"]), row0=196, rows_out=196, transposed=True)
    assert rel_err(q.cpu(), g2["q"][196:]) < 5e-5


def test_sinkhorn_golden(ops, golden):
    g = golden("sinkhorn")
    kat = torch.log(torch.from_numpy(g["kat_in"]).t().contiguous()) * 0.05  # scores whose exp(./eps)^T is the KAT matrix
    assert rel_err(ops.sinkhorn(dev(kat), 3).cpu(), g["kat_it3"]) < 1e-5
    assert rel_err(ops.sinkhorn(dev(kat), 0).cpu(), g["kat_it0"]) < 1e-5
    for tag in "abcd":
        q = ops.sinkhorn(dev(g[f"{tag}_scores"]), int(g[f"{tag}_iters"]))
        assert rel_err(q.cpu(), g[f"{tag}_q"]) < 5e-5, tag
    g2 = golden("sinkhorn_w2")  # global problem solved once, rank 1's rows requested
    q = ops.sinkhorn(dev(g2["scores"]), int(g2["iters"]), row0=196, rows_out=196)
    assert rel_err(q.cpu(), g2["q"][196:]) < 5e-5


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_sinkhorn_c3_global_problem_vs_reference(ops, golden, rank):
    """BASELINE C3's gathered problem on one GPU: K = 200 x 8 ranks x 8320 rows (6272 patches + 2048 queue rows each), solved once over
    the gathered scores; rank r keeps its own first 6272 rows (time_tuning.py:213-215).  Fixture: the reference's my_utils.sinkhorn under
    8 gloo ranks (my_utils.py:250-272), every 52nd row of each rank's q."""
    from timetuning_amd import synth

    g = golden("sinkhorn_w8")
    W, Bl, stride = int(g["world_size"]), int(g["rows_per_rank"]), int(g["stride"])
    gathered = dev(synth.make_sinkhorn_w8_scores())
    assert gathered.shape == (W * Bl, 200)
    q = ops.sinkhorn(gathered, int(g["iters"]), row0=rank * Bl, rows_out=6272).cpu()
    assert q.shape == (6272, 200)
    keep = torch.arange(0, 6272, stride)
    assert rel_err(q[keep], g["q"][rank][: len(keep)]) < 5e-5
    assert rel_err(q.sum(1), torch.ones(6272)) < 1e-5


@pytest.mark.parametrize("W", [2, 8])
def test_sinkhorn_allreduce_form_vs_reference(ops, golden, W):
    """The reference's own distributed pattern (my_utils.py:250-272) on the HIP kernels of one rank's share (tt_sinkhorn_local_begin / step /
    end): W ranks emulated in one process - each keeps its own columns, the K row sums are summed across the ranks between the steps (what
    the all-reduce does) - against the reference's W-rank gloo fixtures (W = 2: the whole q; W = 8: BASELINE C3's 8 x 8320 rows, every
    52nd row), against tt_sinkhorn on the gathered rows, and the plain-C twin of the same three calls."""
    from oracle import cpu_twin
    from timetuning_amd import hip_ops, synth

    if W == 2:
        g = golden("sinkhorn_w2")
        scores = torch.from_numpy(g["scores"]).contiguous()
        iters, Bl, rows_out = int(g["iters"]), scores.shape[0] // 2, scores.shape[0] // 2
        expect = [torch.from_numpy(g["q"][r * Bl:(r + 1) * Bl]) for r in range(W)]
        keep = torch.arange(0, Bl)
    else:
        g = golden("sinkhorn_w8")
        scores = torch.from_numpy(synth.make_sinkhorn_w8_scores())
        iters, Bl, rows_out = int(g["iters"]), int(g["rows_per_rank"]), 6272
        keep = torch.arange(0, rows_out, int(g["stride"]))
        expect = [torch.from_numpy(g["q"][r][: len(keep)]) for r in range(W)]
    K = scores.shape[1]

    def solve(lib, device):
        parts = [scores[r * Bl:(r + 1) * Bl].contiguous().to(device) for r in range(W)]
        sks = [hip_ops.SinkhornLocal(Bl, W * Bl, K, device, lib=lib) for _ in range(W)]
        us = [sk.begin(p_, 0.05) for sk, p_ in zip(sks, parts)]
        for it in range(iters):
            u = torch.stack(us).sum(0)                      # the all-reduce
            if it + 1 < iters:
                us = [sk.step(u) for sk in sks]
        return [sk.end(u, rows_out).cpu() for sk in sks]

    qs = solve(None, "cuda")
    for r in range(W):
        assert rel_err(qs[r][keep], expect[r]) < 5e-5, r
        assert rel_err(qs[r].sum(1), torch.ones(rows_out)) < 1e-5
    q_gathered = ops.sinkhorn(dev(scores), iters, row0=Bl, rows_out=rows_out).cpu()     # rank 1's rows of the all-gather variant
    assert rel_err(qs[1], q_gathered) < 5e-6
    if W == 2:
        qt = solve(cpu_twin.load(), "cpu")
        assert all(rel_err(qs[r], qt[r]) < 5e-6 for r in range(W))
        sk0 = hip_ops.SinkhornLocal(Bl, W * Bl, K, "cuda")                               # zero iterations: column-normalised exp(scores / eps)
        sk0.begin(dev(scores[:Bl]), 0.05)
        e0 = torch.exp(scores[:Bl].double() / 0.05)
        assert rel_err(sk0.end(None).cpu(), e0 / e0.sum(1, keepdim=True)) < 1e-5


def test_sinkhorn_c2_size_vs_oracle(ops):
    """BASELINE C2: K=200, B=6272 (+ queue rows variant), checked against the oracle and by invariants."""
    for B in (6272, 6272 + 2048):
        x = F.normalize(rnd(f"skx{B}", B, 64), dim=1)
        p = F.normalize(rnd("skp", 200, 64), dim=1)
        scores = x @ p.t()
        q = ops.sinkhorn(dev(scores), 10).cpu()
        ref = O.sinkhorn(torch.exp(scores.double() / 0.05).t(), 10)
        assert rel_err(q, ref) < 5e-5
        assert rel_err(q.sum(1), torch.ones(B)) < 1e-5          # rows of q sum to 1
        col = q.double().sum(0)                                   # prototypes are used (nearly) equally
        assert (col.max() / col.min()) < 1.05


@pytest.mark.parametrize("B,K,iters", [(6272, 200, 10), (8320, 200, 10), (50176, 200, 10), (392, 50, 3), (1000, 333, 1), (6272, 200, 0), (777, 512, 5)])
def test_sinkhorn_one_launch_equals_the_launch_per_iteration_path(ops, B, K, iters):
    """sk_persistent_kernel (round 5: the whole solve in ONE launch - E resident in LDS, or in the workspace for problems beyond the chip's LDS
    [50176 rows: the 8-rank global problem]; only the K row sums cross workgroups, through write-through partials + an arrival counter)
    against the launch-per-iteration kernels (knob TT_SK_PERSIST = 0) and the fp64 oracle; run-to-run bit equality (the partials are folded in
    workgroup order whoever arrives first); requested row windows."""
    x = F.normalize(rnd(f"sk1.x{B}.{K}", B, 48), dim=1)
    p = F.normalize(rnd(f"sk1.p{K}", K, 48), dim=1)
    scores = dev((x @ p.t()).contiguous())
    from timetuning_amd import hip_ops
    q0 = ops.sinkhorn(scores, iters)                     # the product path: one launch per iteration
    try:
        hip_ops.set_tuning_knob("TT_SK_PERSIST", 1)      # (measured slower than it: kept behind the knob, see sinkhorn.hip)
        q1 = ops.sinkhorn(scores, iters)
        for _ in range(3):
            assert torch.equal(ops.sinkhorn(scores, iters), q1)
        r0, n = B // 3, B // 4
        qw = ops.sinkhorn(scores, iters, row0=r0, rows_out=n)
        assert torch.equal(qw, q1[r0:r0 + n])
    finally:
        hip_ops.set_tuning_knob("TT_SK_PERSIST", 0)
    assert rel_err(q1.cpu(), q0.cpu()) < 2e-6 and rel_l2(q1.cpu(), q0.cpu()) < 1e-6
    if B <= 8320:
        ref = O.sinkhorn(torch.exp((x @ p.t()).double() / 0.05).t(), iters)
        assert rel_err(q1.cpu(), ref) < 5e-5


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_label_propagation_golden(ops, golden, tag):
    g = golden("label_prop")
    gg, fs, D, K, nlast, r, topk = [int(v) for v in g[f"{tag}_cfg"]]
    feats = torch.from_numpy(g[f"{tag}_feats"])                      # [fs, n, D]
    if D % 4:
        pytest.skip("feature dim not a multiple of 4")
    xn = ops.l2norm_fwd(dev(feats.reshape(-1, D))).reshape(fs, 1, gg * gg, D)
    labels, pmap = ops.label_propagate(xn, dev(g[f"{tag}_q0"]).reshape(1, gg * gg, K), nlast, r, topk, 0.1, return_pmap=True)
    ref = g[f"{tag}_maps"][-1].reshape(K, gg * gg).T                 # [n, K] fp64
    got = pmap.cpu().numpy()[0]
    # a near-tie in the top-k can legitimately swap one source; require agreement on (almost) all queries
    bad = np.abs(got - ref).max(1) > 1e-5 * np.abs(ref).max()
    assert bad.mean() <= 0.01, f"{bad.sum()} of {bad.size} queries differ"
    assert (labels.cpu().numpy()[0] != ref.argmax(1)).mean() <= 0.01


def test_label_propagation_batch(ops):
    bs, fs, g, D, K = 3, 4, 14, 384, 200
    feats = rnd("lpf", fs, bs, g * g, D)
    feats[1:] = 0.7 * feats[:1] + 0.3 * feats[1:]
    q0 = F.softmax(rnd("lpq", bs, g * g, K) * 3, -1)
    xn = ops.l2norm_fwd(dev(feats.reshape(-1, D))).reshape(fs, bs, g * g, D)
    labels, pmap = ops.label_propagate(xn, dev(q0), return_pmap=True)
    for b in range(bs):
        seed = q0[b].view(g, g, K).permute(2, 0, 1).unsqueeze(0)
        ref = O.propagate_labels(7, 6, 5, g, feats[:, b], seed)[-1].reshape(K, g * g).T.numpy()
        bad = np.abs(pmap[b].cpu().numpy() - ref).max(1) > 1e-5 * np.abs(ref).max()
        assert bad.mean() <= 0.01
        assert (labels[b].cpu().numpy() != ref.argmax(1)).mean() <= 0.01


def test_label_propagation_in_two_calls(ops, monkeypatch):
    """tt_label_propagate_sims + tt_label_propagate_from_sims (the training step runs the first on a side stream, beside the Sinkhorn
    solve that produces the seed) launch what tt_label_propagate launches: labels and map bit for bit the one-call form's, also when
    the similarities were computed on another stream; when the similarities do not fit one chunk the first half declines (None)."""
    bs, fs, g, D, K = 3, 4, 14, 384, 200
    feats = rnd("lpf", fs, bs, g * g, D)
    feats[1:] = 0.7 * feats[:1] + 0.3 * feats[1:]
    q0 = dev(F.softmax(rnd("lpq", bs, g * g, K) * 3, -1))
    xn = ops.l2norm_fwd(dev(feats.reshape(-1, D))).reshape(fs, bs, g * g, D)
    for prec in ("f16x3", "f32"):
        ops.set_gemm_precision(prec)
        try:
            labels, pmap = ops.label_propagate(xn, q0, return_pmap=True)
            sims = ops.label_propagate_sims(xn, K)
            assert sims is not None
            l2, p2 = ops.label_propagate(xn, q0, return_pmap=True, sims=sims)
            assert torch.equal(l2, labels) and torch.equal(p2, pmap), prec
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                sims = ops.label_propagate_sims(xn, K)
                ready = torch.cuda.Event()
                ready.record()
            torch.cuda.current_stream().wait_event(ready)
            l3, p3 = ops.label_propagate(xn, q0, return_pmap=True, sims=sims)
            torch.cuda.current_stream().synchronize()
            assert torch.equal(l3, labels) and torch.equal(p3, pmap), prec
        finally:
            ops.set_gemm_precision("f16x3")
    # more than one chunk: 24 x 3 clips x 3 target frames x 196^2 x 4 B = 33 MB of similarities against a 4 MB cap
    xn_w = xn.repeat(1, 24, 1, 1).contiguous()
    monkeypatch.setenv("TT_LP_SIMS_CAP_MB", "4")
    assert ops.label_propagate_sims(xn_w, K) is None
    monkeypatch.delenv("TT_LP_SIMS_CAP_MB")
    assert ops.label_propagate_sims(xn_w, K) is not None


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_davis_protocol_golden(ops, golden, tag):
    """N4: evaluation-protocol propagation (4 context frames, 25x25 window - 16 candidates per thread on the 28x28 grid -,
    all maps returned), fused bilinear upsampling + arg-max, confusion counts / Jaccard; against the reference's outputs."""
    from timetuning_amd import mask_propagation as MP

    d = golden("davis_protocol")
    g_, fs, D, C, R = [int(v) for v in d[f"{tag}_cfg"]]
    feats = dev(d[f"{tag}_feats"])
    ann = dev(d[f"{tag}_annotation"].astype(np.int64))

    class _FE:  # propagate_labels only reads spatial_resolution when the features exist
        spatial_resolution = g_

    maps = MP.propagate_labels(4, 12, 5, _FE(), feats, MP.to_one_hot(ann.unsqueeze(0)).unsqueeze(0), features_exist=True)
    assert len(maps) == fs - 1 and maps[0].shape == (C, g_, g_) and maps[0].dtype == torch.float64
    got = torch.stack(maps).cpu().numpy()
    ref = d[f"{tag}_maps"]
    bad = np.abs(got - ref).max(1) > 1e-5 * np.abs(ref).max()
    assert bad.mean() <= 0.01
    # upsample + argmax on the REFERENCE maps (isolates the fused kernel from top-k near-ties upstream)
    ref_nk = torch.from_numpy(ref).reshape(fs - 1, C, g_ * g_).transpose(1, 2).contiguous()
    pred = ops.upsample_argmax(dev(ref_nk), R).cpu().numpy()
    mism = pred != d[f"{tag}_pred"]
    assert not (mism & ~d[f"{tag}_near_tie"]).any()
    # confusion counts / Jaccard against the oracle
    gt = torch.from_numpy(np.roll(d[f"{tag}_annotation"].astype(np.int64), (2 * (fs - 1) * R // 112, 3 * (fs - 1) * R // 112), (0, 1)))
    last = torch.from_numpy(pred[-1].astype(np.int64))
    counts = ops.confusion_counts(dev(last), dev(gt), C).cpu()
    want = torch.zeros(C, C, dtype=torch.int64)
    want.index_put_((gt.reshape(-1), last.reshape(-1)), torch.ones(R * R, dtype=torch.int64), accumulate=True)
    assert torch.equal(counts, want)
    j, per_class = MP.jaccard(dev(last), dev(gt), C)
    assert abs(j - O.jaccard(last, gt, C)) < 1e-12 and j > 0.5


def test_upsample_argmax_vs_torch(ops):
    """Random fp64 maps, K = 21 channels, 14x14 -> 224x224 and 28x28 -> 100x100 (non-integer scale)."""
    for g_, R, K in ((14, 224, 21), (28, 100, 5), (3, 7, 2)):
        maps = torch.from_numpy(np.random.default_rng(g_).random((3, g_ * g_, K)))
        up = F.interpolate(maps.transpose(1, 2).reshape(3, K, g_, g_), size=(R, R), mode="bilinear", align_corners=False)
        want = up.argmax(1)
        top2 = up.topk(2, dim=1).values
        got = ops.upsample_argmax(dev(maps), R).cpu()
        mism = got != want
        assert not (mism & ((top2[:, 0] - top2[:, 1]) > 1e-12)).any()
        assert mism.float().mean() < 1e-3


def test_ce_loss(ops):
    rows, K = 392, 200
    s, lab = rnd("ces", rows, K) * 0.3, torch.from_numpy(np.random.default_rng(0).integers(0, K, rows))
    sd = s.double().requires_grad_(True)
    ref = F.cross_entropy(sd / 0.1, lab)
    ref.backward()
    loss, ds = ops.ce_loss_fwd_bwd(dev(s), dev(lab), 0.1)
    assert abs(loss.item() - ref.item()) < 1e-5
    assert rel_err(ds.cpu(), sd.grad) < TOL


def test_ce_loss_row_weights(ops):
    """--use_mask loss: CrossEntropyLoss(reduction='none') * mask, mean over all rows (time_tuning.py:226-227,298-300)."""
    rows, K = 392, 50
    s, lab = rnd("cws", rows, K) * 0.3, torch.from_numpy(np.random.default_rng(1).integers(0, K, rows))
    w = torch.from_numpy((np.random.default_rng(2).random(rows) < 0.6).astype(np.float32))
    sd = s.double().requires_grad_(True)
    ref = (F.cross_entropy(sd / 0.1, lab, reduction="none") * w.double()).mean()
    ref.backward()
    loss, ds = ops.ce_loss_fwd_bwd(dev(s), dev(lab), 0.1, row_weight=dev(w))
    assert abs(loss.item() - ref.item()) < 1e-5
    assert rel_err(ds.cpu(), sd.grad) < TOL
    assert (ds.cpu()[w == 0] == 0).all()


def test_scale_rows(ops):
    x, w = rnd("srx", 392, 256), rnd("srw", 392)
    y = ops.scale_rows_(dev(x.clone()), dev(w))
    assert torch.equal(y.cpu(), x * w[:, None])


def _mask_mismatch_excusable(mask, want, margin, tol=2e-6):
    """A pixel may land on the other side of the mass cut only where its cumulative mass is within rounding of the
    cut; such a flip can also change which neighbours form a <= 2-pixel component, so the excuse covers 3x3
    surroundings of any near-cut pixel."""
    mism = mask != want
    if not mism.any():
        return True
    Fr, n = mask.shape
    g = int(round(n ** 0.5))
    near = (margin < tol).reshape(Fr, 1, g, g).float()
    near = F.max_pool2d(near, 5, 1, 2).reshape(Fr, n).bool()   # 2 rings: the flipped pixel's neighbours' neighbours
    return bool((~mism | near).all())


@pytest.mark.parametrize("g_", [14, 28, 7])
def test_foreground_mask_golden(ops, golden, g_):
    """process_attentions (models.py:93-131): exact on the frames the reference itself can process, and against the
    oracle's intended small-component removal on the frames where the reference raises IndexError."""
    d = golden("attention_mask")
    cls = d[f"attn_cls_g{g_}"]
    ok = d[f"ref_ok_g{g_}"].astype(bool)
    mask, blurred, margin = ops.foreground_mask_from_probs(dev(cls), g_, return_aux=True)
    mask, margin = mask.cpu(), margin.cpu()
    Fr, H, N = cls.shape
    attn = torch.zeros(Fr, H, N, N)
    attn[:, :, 0, :] = torch.from_numpy(cls)
    want, want_blur, _ = O.process_attentions(attn, g_, return_blurred=True)
    want = want.reshape(Fr, -1)
    assert rel_err(blurred.cpu(), want_blur) < TOL
    assert _mask_mismatch_excusable(mask, want, margin)
    ref = torch.from_numpy(d[f"mask_g{g_}"]).reshape(Fr, -1)
    assert _mask_mismatch_excusable(mask[ok], ref[ok], margin[ok])
    assert (mask == want).float().mean() > 0.999
    assert (~ok).any() and ok.sum() >= 4  # both kinds of frame are present in the fixture


def test_foreground_mask_small_components(ops):
    """Hand-made thresholded maps: 1- and 2-pixel components (8-connected) go, a 3-pixel chain and blocks stay."""
    g_ = 9
    n = g_ * g_
    keep = np.zeros((g_, g_), np.float32)
    keep[0, 0] = 1                          # single
    keep[0, 4] = keep[1, 5] = 1             # diagonal pair
    keep[8, 0] = keep[8, 1] = 1             # horizontal pair on the border
    keep[3, 0] = keep[4, 0] = keep[5, 1] = 1  # 3-chain with a diagonal link
    keep[4:8, 4:8] = 1                      # block
    # attention whose kept mass is exactly `keep`: big values there, 1e-6 elsewhere, a 1-tap blur (identity) and a mass
    # threshold of 99.9 % so that only the tiny values fall below the cut
    att = np.where(keep.reshape(-1) > 0, 1.0, 1e-6).astype(np.float32)
    att /= att.sum()
    cls = np.zeros((1, 2, n + 1), np.float32)
    cls[0, :, 1:] = att
    mask = ops.foreground_mask_from_probs(dev(cls), g_, threshold=0.999, kernel_size=1).cpu().reshape(g_, g_).numpy()
    want = keep.copy()
    want[0, 0] = want[0, 4] = want[1, 5] = want[8, 0] = want[8, 1] = 0
    assert (mask == want).all()
    lab = O.label_components(keep[None])  # the oracle's component rule agrees on which components are small
    small = [k for k in range(1, lab.max() + 1) if (lab == k).sum() <= 2]
    assert sum((lab == k).sum() for k in small) == 5


def test_foreground_mask_from_qkv(ops):
    """The qkv entry point recomputes the cls-query probabilities: compare with the oracle on attention built by torch
    from the same qkv (F=5 frames, 6 heads x 64, g=14) and with the probs entry point."""
    Fr, H, hd, g_ = 5, 6, 64, 14
    N, D = g_ * g_ + 1, 6 * 64
    qkv = rnd("fmqkv", Fr, N, 3 * D) * 1.5
    # smooth the keys over the grid so that the attention (and the mask) has spatial structure
    kk = qkv[:, 1:, D:2 * D].reshape(Fr, g_, g_, D).permute(0, 3, 1, 2)
    kk = F.avg_pool2d(kk, 5, 1, 2).permute(0, 2, 3, 1).reshape(Fr, N - 1, D)
    qkv[:, 1:, D:2 * D] = 3.0 * kk
    q = qkv[:, :, :D].reshape(Fr, N, H, hd).permute(0, 2, 1, 3).double()
    k = qkv[:, :, D:2 * D].reshape(Fr, N, H, hd).permute(0, 2, 1, 3).double()
    attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1).float()
    want, want_blur, _ = O.process_attentions(attn, g_, return_blurred=True)
    mask, blurred, margin = ops.foreground_mask(dev(qkv), H, g_, return_aux=True)
    assert rel_err(blurred.cpu(), want_blur) < TOL
    assert _mask_mismatch_excusable(mask.cpu(), want.reshape(Fr, -1), margin.cpu())
    assert 0.2 < mask.mean().item() < 0.9
    m2 = ops.foreground_mask_from_probs(dev(attn[:, :, 0, :]), g_)
    assert (m2 == mask).float().mean().item() > 0.995


def test_adamw_ema_queue(ops):
    shapes = [(200, 256), (1024,), (384, 1536), (7,)]
    ps = [torch.nn.Parameter(rnd(f"op{i}", *s)) for i, s in enumerate(shapes)]
    ref_opt = torch.optim.AdamW([{"params": ps[:2], "lr": 1e-3, "weight_decay": 0.04}, {"params": ps[2:], "lr": 1e-4, "weight_decay": 0.0}])
    mine = [dict(p=dev(p.detach().clone()), m=None, v=None) for p in ps]
    for e in mine:
        e["m"], e["v"] = torch.zeros_like(e["p"]), torch.zeros_like(e["p"])
    for step in range(1, 4):
        grads = [rnd(f"og{step}.{i}", *s) for i, s in enumerate(shapes)]
        for p, gr in zip(ps, grads):
            p.grad = gr.clone()
        ref_opt.step()
        ents = [(e["p"], dev(gr), e["m"], e["v"], 1e-3 if i < 2 else 1e-4, 0.04 if i < 2 else 0.0) for i, (e, gr) in enumerate(zip(mine, grads))]
        ops.adamw_step_(ents, step)
        for e, p in zip(mine, ps):
            assert rel_err(e["p"].cpu(), p.detach()) < 1e-6
    t, s = rnd("et", 100003), rnd("es", 100003)
    m = 0.9951234567
    tt = ops.ema_update_(dev(t), dev(s), m)
    assert rel_err(tt.cpu(), t * (1.0 - m) + s * m) < 1e-6
    queue, feats = rnd("qq", 40, 32), rnd("qf", 392, 32)
    idx = torch.randperm(392)[:20]
    ref_q = queue.clone()
    ref_q[20:] = ref_q[:-20].clone()
    ref_q[:20] = feats[idx]
    qd = ops.queue_push_(dev(queue), dev(feats), dev(idx))
    assert torch.equal(qd.cpu(), ref_q)


@pytest.mark.parametrize("M,N,K", [(1280, 384, 384), (6272, 1024, 1024), (640, 1536, 384)])
def test_linear_precision_modes(ops, M, N, K):
    """Opt-in bf16 MFMA instances of the forward Linear: "bf16x3" (split precision) and "bf16" (BASELINE C4), against fp64.
    The default f32 mode is restored afterwards; every other test in this suite runs in f32."""
    x, w, b, r = rnd("px", M, K), rnd("pw", N, K, std=0.05), rnd("pb", N), rnd("pr", M, N)
    ref = F.gelu(F.linear(x.double(), w.double(), b.double())) + r.double()
    errs = {}
    try:
        for mode in ("f32", "bf16x3", "bf16"):
            ops.set_gemm_precision(mode)
            assert ops.get_gemm_precision() == mode
            y = ops.linear_fwd(dev(x), dev(w), dev(b), residual=dev(r), act=1)
            errs[mode] = rel_err(y.cpu(), ref)
    finally:
        ops.set_gemm_precision("f32")
    assert errs["f32"] < TOL
    assert errs["bf16x3"] < 1e-4, errs     # ~2^-16 per product, averaged over K
    assert errs["f32"] < errs["bf16x3"] < errs["bf16"] < 2e-2, errs


def test_label_propagation_with_exact_ties(ops):
    """identical tokens (flat frames) tie EXACTLY, so "top-k plus ties" keeps every windowed source; the propagated
    map must still be normalised (columns of aff sum to 1 -> each query's map sums to the seed's row sum) and equal the oracle."""
    from oracle import timet_oracle as O

    fs, bs, g, D, K = 3, 1, 14, 32, 6
    n = g * g
    tok = torch.ones(fs, bs, n, D) / D ** 0.5                                   # every token identical, unit norm
    seed = torch.softmax(torch.from_numpy(synth.normal("ties.seed", (bs, n, K))) * 2, dim=-1)
    labels, pmap = ops.label_propagate(dev(tok), dev(seed.float()), 7, 6, 5, 0.1, return_pmap=True)
    assert torch.allclose(pmap.sum(-1).cpu(), torch.ones(bs, n, dtype=torch.float64), atol=1e-5)
    omaps = O.propagate_labels(7, 6, 5, g, tok[:, 0].clone(), seed[0].view(g, g, K).permute(2, 0, 1).unsqueeze(0))
    want = omaps[-1].reshape(K, n).t()
    assert rel_err(pmap[0].cpu(), want) < 1e-5


def test_label_propagation_chunked_similarities(ops, monkeypatch):
    """The target x context similarities are computed up front for as many target frames as fit a workspace cap: a 10-frame clip
    with 3 context frames (both slot regimes: queue filling, queue full) run whole and in chunks of 1 / 2 / 4 target frames must
    give the same maps bit for bit, and those must equal the oracle's."""
    bs, fs, g, D, K, nlast = 2, 10, 7, 32, 5, 3
    n = g * g
    feats = rnd("lpc", fs, bs, n, D)
    for t in range(1, fs):
        feats[t] = 0.8 * feats[t - 1] + 0.2 * feats[t]
    q0 = F.softmax(rnd("lpcq", bs, n, K) * 3, -1)
    xn = ops.l2norm_fwd(dev(feats.reshape(-1, D))).reshape(fs, bs, n, D)
    whole = ops.label_propagate_maps(xn, dev(q0), nlast, 2, 3, 0.1)
    assert whole.shape == (fs - 1, bs, n, K)
    for b in range(bs):
        seed = q0[b].view(g, g, K).permute(2, 0, 1).unsqueeze(0)
        ref = torch.stack(O.propagate_labels(nlast, 2, 3, g, feats[:, b], seed)).reshape(fs - 1, K, n).transpose(1, 2).numpy()
        bad = np.abs(whole[:, b].cpu().numpy() - ref).max(-1) > 1e-5 * np.abs(ref).max()
        assert bad.mean() <= 0.01
    # chunked: the cap is in MB, so widen the batch with copies until a frame's similarities exceed one
    reps = 24                                                   # per-frame similarities: 24 x 2 x 4 x 49^2 x 4 B = 1.84 MB
    xn_w, q0_w = xn.repeat(1, reps, 1, 1).contiguous(), dev(q0).repeat(reps, 1, 1).contiguous()
    base = ops.label_propagate_maps(xn_w, q0_w, nlast, 2, 3, 0.1)
    assert torch.equal(base[:, :bs], whole)
    for cap_mb in (1, 4, 8):                                    # 1 (floor), 2 and 4 target frames per chunk
        monkeypatch.setenv("TT_LP_SIMS_CAP_MB", str(cap_mb))
        assert torch.equal(ops.label_propagate_maps(xn_w, q0_w, nlast, 2, 3, 0.1), base), cap_mb
    monkeypatch.delenv("TT_LP_SIMS_CAP_MB")
