#!/usr/bin/env python3
"""Randomised shape fuzz of the HIP ops against torch fp64 / the oracle (development aid; run on the GPU box).
usage: fuzz_ops.py [rounds=40] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from timetuning_amd import hip_ops as ops
from oracle import timet_oracle as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: torch.as_tensor(a).cuda()
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-30))
worst = {}
def note(name, err, tol, info):
    worst[name] = max(worst.get(name, 0.0), err)
    assert err < tol, (name, err, info)

for it in range(rounds):
    # Linear forward: ragged M, N multiple of 64 or not, K multiple of 16 or not, epilogues
    M, N, K = int(rng.integers(1, 700)), int(rng.choice([64, 128, 192, 200, 384, 50])), int(rng.choice([16, 48, 64, 100, 384]))
    x, w, b = torch.randn(M, K), torch.randn(N, K) * 0.1, torch.randn(N)
    res = torch.randn(M, N) if rng.random() < 0.5 else None
    act = int(rng.random() < 0.5)
    ref = F.linear(x.double(), w.double(), b.double())
    if act: ref = F.gelu(ref)
    if res is not None: ref = ref + res.double()
    y = ops.linear_fwd(dev(x), dev(w), dev(b), residual=dev(res) if res is not None else None, act=act)
    note("linear_fwd", rel(y, ref), 3e-5, (M, N, K, act))
    dy = torch.randn(M, N)
    dx = ops.linear_bwd_data(dev(dy), dev(w))
    note("linear_bwd_data", rel(dx, dy.double() @ w.double()), 3e-5, (M, N, K))
    dw, db = ops.linear_bwd_weight(dev(dy), dev(x))
    note("linear_bwd_weight", max(rel(dw, dy.double().t() @ x.double()), rel(db, dy.double().sum(0))), 3e-5, (M, N, K))
    # LayerNorm forward / backward, plain and with the dropped first token
    D = int(rng.choice([64, 128, 256, 384, 512, 768, 1000, 1024]))
    Fr, Nt = int(rng.integers(1, 9)), int(rng.integers(2, 40))
    xx, g, bb = torch.randn(Fr, Nt, D) * 2 + 0.3, 1 + 0.1 * torch.randn(D), 0.1 * torch.randn(D)
    for drop in (False, True):
        xd = xx.double().requires_grad_(True)
        r = F.layer_norm(xd, (D,), g.double(), bb.double(), 1e-6)
        r = r[:, 1:] if drop else r
        dyl = torch.randn_like(r)
        r.backward(dyl)
        y, mean, rstd = ops.layernorm_fwd(dev(xx), dev(g), dev(bb), save_stats=True, drop_first_token=drop)
        note("layernorm_fwd", rel(y.reshape(r.shape), r.detach()), 3e-5, (Fr, Nt, D, drop))
        dxl, dg, dbt = ops.layernorm_bwd(dev(dyl.reshape(-1, D).float()), dev(xx), dev(g), mean, rstd, drop_first_token=drop)
        note("layernorm_bwd", rel(dxl.reshape(xx.shape), xd.grad), 1e-4, (Fr, Nt, D, drop))
        rows = r.reshape(-1, D).shape[0]
        gref = (dyl.reshape(-1, D) * ((xx[:, 1:] if drop else xx).reshape(-1, D).double() - torch.as_tensor(mean.cpu().double())[:, None]) * rstd.cpu().double()[:, None]).sum(0)
        note("layernorm_bwd.dgamma", rel(dg, gref), 1e-4, (Fr, Nt, D, drop, rows))
    # attention
    Fa, Na, H = int(rng.integers(1, 4)), int(rng.choice([5, 17, 50, 197, 256, 257, 300])), int(rng.integers(1, 4))
    qkv = torch.randn(Fa, Na, 3 * H * 64)
    q, k, v = qkv.double().view(Fa, Na, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(Fa, Na, H * 64)
    out, _, _ = ops.attention_fwd(dev(qkv), H)
    note("attention_fwd", rel(out, ref), 3e-5, (Fa, Na, H))
    # label propagation: random grid, window, contexts, prototypes
    gl, Dl, Kl = int(rng.choice([5, 7, 14])), int(rng.choice([16, 32])), int(rng.choice([3, 20, 200, 300]))
    fs, bs, nlast, rad, topk = int(rng.integers(2, 7)), int(rng.integers(1, 4)), int(rng.integers(1, 5)), int(rng.integers(1, 8)), int(rng.integers(1, 7))   # (n_last_frames = 0 crashes the reference itself, mask_propagation.py:489)
    nl = gl * gl
    feats = torch.randn(fs, bs, nl, Dl)
    for t in range(1, fs): feats[t] = 0.7 * feats[t - 1] + 0.3 * feats[t]
    xn = F.normalize(feats, dim=-1)
    seg0 = torch.softmax(torch.randn(bs, nl, Kl) * 2, -1)
    maps = ops.label_propagate_maps(dev(xn), dev(seg0), nlast, rad, topk, 0.1).cpu().numpy()
    bad = tot = 0
    for b_ in range(bs):
        seed = seg0[b_].view(gl, gl, Kl).permute(2, 0, 1).unsqueeze(0)
        refm = torch.stack(O.propagate_labels(nlast, rad, topk, gl, xn[:, b_], seed)).reshape(fs - 1, Kl, nl).transpose(1, 2).numpy()
        d = np.abs(maps[:, b_] - refm).max(-1) > 1e-5 * np.abs(refm).max()
        bad += d.sum(); tot += d.size
    note("label_propagate_maps (fraction of queries off)", bad / tot, 0.03, (gl, Dl, Kl, fs, bs, nlast, rad, topk))
    # patch embedding (lean gather instance for 16-pixel patches)
    Dp, Hh, Ww = int(rng.choice([64, 128, 384])), 16 * int(rng.integers(1, 5)), 16 * int(rng.integers(1, 5))
    nsrc = int(rng.integers(1, 5)); fmap = rng.integers(0, nsrc, int(rng.integers(1, 6))).astype(np.int32)
    img, wp, bp, cls = torch.randn(nsrc, 3, Hh, Ww), torch.randn(Dp, 768) * 0.05, torch.randn(Dp), torch.randn(Dp)
    npat = (Hh // 16) * (Ww // 16)
    pos = torch.randn(1 + npat, Dp)
    conv = F.conv2d(img[torch.as_tensor(fmap).long()].double(), wp.double().view(Dp, 3, 16, 16), bp.double(), stride=16)
    ref = torch.cat([cls.double().expand(len(fmap), 1, Dp), conv.flatten(2).transpose(1, 2)], 1) + pos.double()
    tok = ops.patch_embed_fwd(dev(img), dev(wp), dev(bp), dev(cls), dev(pos), 16, dev(fmap))
    note("patch_embed_fwd", rel(tok, ref), 3e-5, (Dp, Hh, Ww, nsrc, len(fmap)))
    # bf16-plane Linear (P = 1, 3): ragged M, N and K multiples of 64, epilogue variants
    P_ = int(rng.choice([1, 3])); Mp, Np, Kp = int(rng.integers(1, 600)), 64 * int(rng.integers(1, 6)), 64 * int(rng.integers(1, 5))
    xs, wsn, bs_ = torch.randn(Mp, Kp), torch.randn(Np, Kp) * 0.1, torch.randn(Np)
    xp, wpl = ops.split_planes(dev(xs), P_), ops.split_planes(dev(wsn), P_)
    actp = int(rng.random() < 0.5)
    resp = torch.randn(Mp, Np) if rng.random() < 0.5 else None
    out = ops.linear_fwd_planes(xp, wpl, dev(bs_), residual=dev(resp) if resp is not None else None, act=actp)["y"]
    xd_, wd_ = xp.double().sum(0).cpu() if P_ == 3 else xp[0].double().cpu(), wpl.double().sum(0).cpu() if P_ == 3 else wpl[0].double().cpu()
    refp = F.linear(xd_, wd_, bs_.double())
    if actp: refp = F.gelu(refp)
    if resp is not None: refp = refp + resp.double()
    note(f"linear_fwd_planes P={P_}", rel(out, refp), 3e-5, (P_, Mp, Np, Kp, actp))
    # patch embedding on bf16 operands (round 3): against the fp64 conv of the SAME rounded operands
    Dq = int(rng.choice([64, 128, 256, 384, 768]))
    wq = torch.randn(Dq, 768) * 0.05; bq, clsq, posq = torch.randn(Dq), torch.randn(Dq), torch.randn(1 + npat, Dq)
    convq = F.conv2d(img[torch.as_tensor(fmap).long()].to(torch.bfloat16).double(), wq.to(torch.bfloat16).double().view(Dq, 3, 16, 16), bq.double(), stride=16)
    refq = torch.cat([clsq.double().expand(len(fmap), 1, Dq), convq.flatten(2).transpose(1, 2)], 1) + posq.double()
    tokq = ops.patch_embed_fwd_planes(dev(img), ops.split_planes(dev(wq), 1), dev(bq), dev(clsq), dev(posq), 16, dev(fmap))
    note("patch_embed_fwd_planes", rel(tokq, refq), 3e-5, (Dq, Hh, Ww, nsrc, len(fmap)))
    # bf16 attention forward: any token count up to 256, any head count
    Fa, Na, Ha = int(rng.integers(1, 4)), int(rng.integers(1, 257)), int(rng.integers(1, 5))
    qb = (torch.randn(Fa, Na, 3 * Ha * 64) * 0.7).to(torch.bfloat16)
    q_, k_, v_ = qb.double().view(Fa, Na, 3, Ha, 64).permute(2, 0, 3, 1, 4)
    refa = (torch.softmax(q_ @ k_.transpose(-1, -2) * 0.125, -1) @ v_).permute(0, 2, 1, 3).reshape(Fa, Na, Ha * 64)
    note("attention_fwd_bf16", rel(ops.attention_fwd_bf16(dev(qb), Ha).float(), refa), 2e-2, (Fa, Na, Ha))
    # the persistent 8-phase plane GEMM (large ragged M, whole 256 / 128-wide column tiles, every tile-count regime by chance)
    if it % 4 == 0:
        P8 = int(rng.choice([1, 3])); BN = 256 if P8 == 1 else 128; BKq = 128 if P8 == 1 else 64
        M8, N8, K8 = int(rng.integers(9000, 40000)), BN * int(rng.integers(1, 7)), BKq * int(rng.integers(1, 5))
        x8, w8, b8 = torch.randn(M8, K8), torch.randn(N8, K8) * 0.1, torch.randn(N8)
        xp8, wp8 = ops.split_planes(dev(x8), P8), ops.split_planes(dev(w8), P8)
        res8 = torch.randn(M8, N8) if rng.random() < 0.5 else None
        out8 = ops.linear_fwd_planes(xp8, wp8, dev(b8), residual=dev(res8) if res8 is not None else None)["y"]
        ref8 = F.linear(xp8.double().sum(0).cpu(), wp8.double().sum(0).cpu(), b8.double())
        if res8 is not None: ref8 = ref8 + res8.double()
        note(f"linear_fwd_planes (large M) P={P8}", rel(out8, ref8), 3e-5, (P8, M8, N8, K8, res8 is not None))
    # ---- round 4: the fp16-pair kernels ("f16x3")
    # general pair kernel: ragged M, N multiple of 64, K multiple of 32, every output combination
    Mq, Nq, Kq = int(rng.integers(1, 900)), 64 * int(rng.integers(1, 7)), 32 * int(rng.integers(1, 13))
    xq, wq_, bq_ = torch.randn(Mq, Kq), torch.randn(Nq, Kq) * 0.1, torch.randn(Nq)
    xpq, wpq = ops.split_pairs(dev(xq)), ops.split_pairs(dev(wq_))
    resq = torch.randn(Mq, Nq) if rng.random() < 0.5 else None
    actq = int(resq is None and rng.random() < 0.5)
    refq2 = F.linear(xq.double(), wq_.double(), bq_.double())
    oq = ops.linear_fwd_pairs(xpq, wpq, dev(bq_), residual=dev(resq) if resq is not None else None, act=actq, out_f32=True, out_pairs=(resq is None),
                              save_pre=bool(actq))
    refy = (F.gelu(refq2) if actq else refq2) + (resq.double() if resq is not None else 0)
    note("linear_fwd_pairs (general)", rel(oq["y"], refy), 2e-6, (Mq, Nq, Kq, actq, resq is not None))
    if oq["pairs"] is not None: note("linear_fwd_pairs pairs out", rel(ops.join_pairs(oq["pairs"]), refy), 2e-6, (Mq, Nq, Kq, actq))
    if actq: note("linear_fwd_pairs pre", rel(oq["pre"], refq2), 2e-6, (Mq, Nq, Kq))
    # pair attention: any token count (resident kernel up to 256, KV-tiled beyond and forced), any head count
    Fp, Np, Hp = int(rng.integers(1, 4)), int(rng.integers(1, 900)), int(rng.integers(1, 4))
    qf = torch.randn(Fp, Np, 3 * Hp * 64) * 0.8
    qpp = ops.split_pairs(dev(qf.view(Fp * Np, -1))).view(Fp, Np, -1)
    q_, k_, v_ = qf.double().view(Fp, Np, 3, Hp, 64).permute(2, 0, 3, 1, 4)
    scp = q_ @ k_.transpose(-1, -2) * 0.125
    refap = (torch.softmax(scp, -1) @ v_).permute(0, 2, 1, 3).reshape(Fp, Np, Hp * 64)
    for forced in ((0, 1) if Np <= 256 else (0,)):
        ops.set_tuning_knob("TT_ATTN_PAIRS_FLASH", forced)
        op_, of_, lse_ = ops.attention_fwd_pairs(qpp, Hp, out_pairs=True, out_f32=True, save_lse=True)
        ops.set_tuning_knob("TT_ATTN_PAIRS_FLASH", 0)
        note(f"attention_fwd_pairs ({'KV-tiled' if (forced or Np > 256) else 'resident'})", max(rel(of_, refap), rel(ops.join_pairs(op_.view(Fp * Np, -1)).view(Fp, Np, -1), refap)),
             3e-6, (Fp, Np, Hp))
        note("attention_fwd_pairs lse", rel(lse_, torch.logsumexp(scp, -1)), 1e-5, (Fp, Np, Hp))
    # weight gradient from row pairs (transposing LDS reads): any M, N and K multiples of 128
    Mt, Nt_, Kt = int(rng.integers(1, 9000)), 128 * int(rng.integers(1, 5)), 128 * int(rng.integers(1, 5))
    dyt, xt = torch.randn(Mt, Nt_) * 0.05, torch.randn(Mt, Kt)
    dwt = ops.linear_bwd_weight_pairs_tn(ops.split_pairs(dev(dyt)), ops.split_pairs(dev(xt)))
    note("linear_bwd_weight_pairs_tn", rel(dwt, dyt.double().t() @ xt.double()), 2e-6, (Mt, Nt_, Kt))
    if it % 4 == 0:
        # the persistent pair GEMM: large ragged M, every tile-count regime (round-robin, half tiles, K-split) by chance
        M8, N8, K8 = int(rng.integers(9000, 40000)), 128 * int(rng.integers(1, 13)), 96 * int(rng.integers(1, 17))
        x8, w8, b8 = torch.randn(M8, K8), torch.randn(N8, K8) * 0.1, torch.randn(N8)
        res8 = torch.randn(M8, N8) if rng.random() < 0.5 else None
        o8 = ops.linear_fwd_pairs(ops.split_pairs(dev(x8)), ops.split_pairs(dev(w8)), dev(b8), residual=dev(res8) if res8 is not None else None)["y"]
        ref8 = F.linear(x8.double(), w8.double(), b8.double()) + (res8.double() if res8 is not None else 0)
        route8 = ops._lib.load().tt_linear_fwd_pairs_route(M8, N8, K8, 0, 1, int(res8 is not None), 1, 0, 0)
        note(f"linear_fwd_pairs (large M, route {route8})", rel(o8, ref8), 2e-6, (M8, N8, K8, res8 is not None))
print("fuzz ok:", {k: f"{v:.2e}" for k, v in worst.items()})
