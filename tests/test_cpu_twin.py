"""oracle/tt_cpu.c - the plain-C twins of a subset of the C ABI - against the NumPy / torch oracles and golden vectors (CPU),
and the HIP library against the twins through ONE call site with identical prototypes (GPU)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_twin, image_ops as I, timet_oracle as O


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@pytest.fixture(scope="module")
def twin():
    return cpu_twin.load()


def test_twin_exports_and_prototypes(twin):
    from timetuning_amd._lib import SIGNATURES

    for name in cpu_twin.TWINS:
        fn = getattr(twin, "tt_cpu_" + name)
        assert fn.argtypes == SIGNATURES["tt_" + name][1]


def test_twin_sinkhorn_golden(twin, golden):
    g = golden("sinkhorn")
    for tag in "abcd":
        sc = np.ascontiguousarray(g[f"{tag}_scores"], np.float32)
        B, K = sc.shape
        q = np.empty_like(sc)
        assert twin.tt_cpu_sinkhorn(ptr(sc), ptr(q), B, K, 0, B, 0.05, int(g[f"{tag}_iters"]), None, 0, None) == 0
        assert np.abs(q - g[f"{tag}_q"]).max() < 2e-6 * np.abs(g[f"{tag}_q"]).max() + 1e-7, tag
    kat = np.ascontiguousarray((np.log(g["kat_in"].T) * 0.05), np.float32)   # exp(./eps)^T is the KAT matrix
    q = np.empty_like(kat)
    twin.tt_cpu_sinkhorn(ptr(kat), ptr(q), kat.shape[0], kat.shape[1], 0, kat.shape[0], 0.05, 3, None, 0, None)
    assert np.abs(q - g["kat_it3"]).max() < 1e-6


def test_twin_ce(twin):
    rng = np.random.default_rng(0)
    rows, K = 97, 50
    s = (rng.standard_normal((rows, K)) * 0.3).astype(np.float32)
    lab = rng.integers(0, K, rows).astype(np.int64)
    w = (rng.random(rows) < 0.6).astype(np.float32)
    sd = torch.from_numpy(s).double().requires_grad_(True)
    ref = (F.cross_entropy(sd / 0.1, torch.from_numpy(lab), reduction="none") * torch.from_numpy(w).double()).mean()
    ref.backward()
    loss, ds = np.empty(1, np.float32), np.empty_like(s)
    assert twin.tt_cpu_ce_loss_fwd_bwd(ptr(s), ptr(lab), ptr(w), ptr(loss), ptr(ds), rows, K, 0.1, None, 0, None) == 0
    assert abs(loss[0] - ref.item()) < 1e-5 and np.abs(ds - sd.grad.numpy()).max() < 1e-6


def _resize_with(lib, prefix, frames, oh, ow, dev=None):
    """Image.resize((ow, oh)) of uint8 [F,H,W,3] through <prefix>img_resample_h/_v with host (twin) or device (HIP) buffers."""
    Fr, H, W, _ = frames.shape
    kh, bh = I.resample_coeffs(W, ow)
    kv, bv = I.resample_coeffs(H, oh)
    if dev is None:
        mid = np.empty((Fr, H, ow, 3), np.uint8)
        out = np.empty((Fr, oh, ow, 3), np.uint8)
        getattr(lib, prefix + "img_resample_h")(ptr(frames), ptr(mid), ptr(kh), ptr(bh), Fr, H, W, 0, 0, H, ow, kh.shape[1], None)
        getattr(lib, prefix + "img_resample_v")(ptr(mid), ptr(out), None, ptr(kv), ptr(bv), Fr, H, ow, 0, oh, kv.shape[1], 0, None, None, None)
        return out
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fr, khd, bhd, kvd, bvd = t(frames), t(kh), t(bh), t(kv), t(bv)
    mid = torch.empty((Fr, H, ow, 3), dtype=torch.uint8, device=dev)
    out = torch.empty((Fr, oh, ow, 3), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert getattr(lib, prefix + "img_resample_h")(fr.data_ptr(), mid.data_ptr(), khd.data_ptr(), bhd.data_ptr(), Fr, H, W, 0, 0, H, ow, kh.shape[1], st) == 0
    assert getattr(lib, prefix + "img_resample_v")(mid.data_ptr(), out.data_ptr(), None, kvd.data_ptr(), bvd.data_ptr(), Fr, H, ow, 0, oh, kv.shape[1], 0,
                                                  None, None, st) == 0
    return out.cpu().numpy()


def test_twin_image_ops_bit_exact(twin):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (2, 45, 61, 3), dtype=np.uint8)
    got = _resize_with(twin, "tt_cpu_", a, 32, 40)
    assert all((got[f] == I.resize_bilinear(a[f], (40, 32))).all() for f in range(2))
    for mode, fn, arg in ((0, lambda x: I.gray3(x), 1.0), (1, lambda x: I.enhance_brightness(x, 1.3), 1.3), (2, lambda x: I.enhance_contrast(x, 0.4), 0.4),
                          (3, lambda x: I.enhance_saturation(x, 1.7), 1.7)):
        b = a.copy()
        assert twin.tt_cpu_img_color(ptr(b), 2, 45, 61, mode, arg, 0, None, None) == 0
        assert (b == np.stack([fn(x) for x in a])).all(), mode
    b = a.copy()
    twin.tt_cpu_img_color(ptr(b), 2, 45, 61, 4, 1.0, I.hue_shift_u8(-0.13), None, None)
    assert (b == np.stack([I.adjust_hue(x, -0.13) for x in a])).all()
    r, ww, fw = I.box_weights(I.gaussian_box_radius(1.3))
    cur = a
    for direction in (0, 1):
        for _ in range(3):
            nxt = np.empty_like(cur)
            twin.tt_cpu_img_box_blur(ptr(cur), ptr(nxt), 2, 45, 61, direction, r, ww, fw, None)
            cur = nxt
    assert (cur == np.stack([I.gaussian_blur(x, 1.3) for x in a])).all()


def test_twin_misc(twin):
    rng = np.random.default_rng(2)
    pred, gt = rng.integers(0, 7, 5000).astype(np.int64), rng.integers(0, 7, 5000).astype(np.int64)
    counts = np.empty((7, 7), np.uint64)
    twin.tt_cpu_confusion_counts(ptr(pred), ptr(gt), 5000, 7, ptr(counts), None)
    want = np.zeros((7, 7), np.int64)
    np.add.at(want, (gt, pred), 1)
    assert (counts.astype(np.int64) == want).all()
    maps = rng.random((2, 49, 5))
    out = np.empty((2, 20, 20), np.int64)
    twin.tt_cpu_upsample_argmax(ptr(maps), ptr(out), 2, 7, 5, 20, None)
    up = F.interpolate(torch.from_numpy(maps).transpose(1, 2).reshape(2, 5, 7, 7), size=(20, 20), mode="bilinear", align_corners=False)
    assert (out == up.argmax(1).numpy()).mean() > 0.999
    x, c = rng.standard_normal((300, 9)).astype(np.float32), rng.standard_normal((4, 9)).astype(np.float32)
    lab, d2 = np.empty(300, np.int32), np.empty(300, np.float32)
    twin.tt_cpu_kmeans_assign(ptr(x), ptr(c), ptr(lab), ptr(d2), 300, 9, 4, None)
    ref = ((x[:, None].astype(np.float64) - c[None]) ** 2).sum(-1)
    assert (lab == ref.argmin(1)).all() and np.abs(d2 - ref.min(1)).max() < 1e-4
    mean, var = np.empty(9), np.empty(9)
    twin.tt_cpu_col_moments(ptr(x), ptr(mean), ptr(var), 300, 9, None, 0, None)
    assert np.abs(mean - x.astype(np.float64).mean(0)).max() < 1e-12 and np.abs(var - x.astype(np.float64).var(0)).max() < 1e-12


@pytest.mark.gpu
def test_hip_library_equals_its_cpu_twin(twin):
    """One call site, two libraries: the HIP entry points on device buffers and their C twins on host buffers."""
    from timetuning_amd import _lib

    hip = _lib.load()
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (3, 70, 50, 3), dtype=np.uint8)
    assert (_resize_with(hip, "tt_", a, 33, 64, dev="cuda") == _resize_with(twin, "tt_cpu_", a, 33, 64)).all()
    st = torch.cuda.current_stream().cuda_stream
    for mode, factor, shift in ((0, 1.0, 0), (1, 0.7, 0), (2, 1.6, 0), (3, 0.2, 0), (4, 1.0, 201)):
        d = torch.from_numpy(a.copy()).cuda()
        ws = torch.empty(3, dtype=torch.int64, device="cuda")
        assert hip.tt_img_color(d.data_ptr(), 3, 70, 50, mode, factor, shift, ws.data_ptr(), st) == 0
        b = a.copy()
        twin.tt_cpu_img_color(ptr(b), 3, 70, 50, mode, factor, shift, None, None)
        assert (d.cpu().numpy() == b).all(), mode
    sc = (rng.standard_normal((392, 50)) * 0.2).astype(np.float32)
    q_cpu = np.empty_like(sc)
    twin.tt_cpu_sinkhorn(ptr(sc), ptr(q_cpu), 392, 50, 0, 392, 0.05, 10, None, 0, None)
    d_sc, d_q = torch.from_numpy(sc).cuda(), torch.empty(392, 50, device="cuda")
    nb = hip.tt_sinkhorn_workspace_bytes(392, 50)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    assert hip.tt_sinkhorn(d_sc.data_ptr(), d_q.data_ptr(), 392, 50, 0, 392, 0.05, 10, ws.data_ptr(), nb, st) == 0
    assert np.abs(d_q.cpu().numpy() - q_cpu).max() < 5e-5 * q_cpu.max()


# ---- round 2: the hot path's row ops and (naive) matrix products ------------------------------------------------------------

class _Side:
    """One call site for both libraries: ``run(name, *args)`` calls ``<prefix><name>`` with numpy arrays passed as host pointers
    (twin) or as device copies (HIP); arrays listed in ``outs`` are copied back after the call."""

    def __init__(self, lib, prefix, device=None):
        self.lib, self.prefix, self.device = lib, prefix, device

    def run(self, name, *args, outs=()):
        fn = getattr(self.lib, self.prefix + name)
        if self.device is None:
            rc = fn(*[ptr(a) if isinstance(a, np.ndarray) else a for a in args])
            assert rc == 0, (name, rc)
            return
        dev = {}
        conv = []
        for a in args:
            if isinstance(a, np.ndarray):
                t = torch.from_numpy(a).to(self.device)
                dev[id(a)] = (a, t)
                conv.append(t.data_ptr())
            elif a == "STREAM":
                conv.append(torch.cuda.current_stream().cuda_stream)
            else:
                conv.append(a)
        rc = fn(*conv)
        assert rc == 0, (name, rc, self.lib.tt_last_error() if hasattr(self.lib, "tt_last_error") else "")
        torch.cuda.synchronize()
        for o in outs:
            o[...] = dev[id(o)][1].cpu().numpy()


def _round2_cases(side, tol):
    """Runs every round-2 twin op on ``side`` and returns the results; the caller compares them with a reference."""
    st = "STREAM" if side.device else None
    rng = np.random.default_rng(7)
    f32 = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)
    R = {}
    # Linear forward / backward
    M, N, K = 70, 64, 48
    x, w, b, res = f32(M, K), f32(N, K, scale=0.1), f32(N), f32(M, N)
    y, pre = np.empty((M, N), np.float32), np.empty((M, N), np.float32)
    side.run("linear_fwd", x, w, b, res, y, pre, M, N, K, 1, 0, st, outs=(y, pre))
    R["linear_fwd"] = (x, w, b, res, y.copy(), pre.copy())
    dy, gp = f32(M, N), f32(M, K)
    dx = np.empty((M, K), np.float32)
    side.run("linear_bwd_data", dy, w, gp, dx, M, N, K, st, outs=(dx,))
    dw, db = np.empty((N, K), np.float32), np.empty(N, np.float32)
    nb = 1 << 22
    ws = np.empty(nb, np.uint8)
    side.run("linear_bwd_weight", dy, x, dw, db, M, N, K, ws, nb, st, outs=(dw, db))
    R["linear_bwd"] = (dy, gp, dx.copy(), dw.copy(), db.copy())
    # LayerNorm (with and without the cls-dropping row map), l2norm, prototype renormalisation
    Fr, Nt, D = 3, 5, 40
    xl, g, be = f32(Fr, Nt, D), 1 + 0.1 * f32(D), 0.1 * f32(D)
    yl, mu, rs = np.empty((Fr * Nt, D), np.float32), np.empty(Fr * Nt, np.float32), np.empty(Fr * Nt, np.float32)
    side.run("layernorm_fwd", xl, g, be, yl, mu, rs, Fr * Nt, D, 1e-6, 0, st, outs=(yl, mu, rs))
    yd = np.empty((Fr * (Nt - 1), D), np.float32)
    side.run("layernorm_fwd", xl, g, be, yd, None, None, Fr * (Nt - 1), D, 1e-6, Nt, st, outs=(yd,))
    R["layernorm"] = (xl, g, be, yl.copy(), mu.copy(), rs.copy(), yd.copy())
    xn, inv = np.empty((M, K), np.float32), np.empty(M, np.float32)
    side.run("l2norm_fwd", x, K, xn, inv, M, K, st, outs=(xn, inv))
    wn = w.copy()
    side.run("normalize_rows_inplace", wn, N, K, st, outs=(wn,))
    R["l2norm"] = (xn.copy(), inv.copy(), wn.copy())
    # attention forward: out, lse and probabilities
    Fa, Na, H = 2, 37, 2
    qkv = f32(Fa, Na, 3 * H * 64, scale=0.5)
    out, lse, probs = np.empty((Fa, Na, H * 64), np.float32), np.empty((Fa, H, Na), np.float32), np.empty((Fa, H, Na, Na), np.float32)
    side.run("attention_fwd", qkv, out, lse, probs, Fa, Na, H, 64, 0.125, st, outs=(out, lse, probs))
    R["attention"] = (qkv, out.copy(), lse.copy(), probs.copy())
    # EMA, queue push, row scaling, sinkhorn on the positive matrix, bf16 planes, mismatch count
    t, s = f32(1000), f32(1000)
    t2 = t.copy()
    side.run("ema_update", t2, s, 1000, 0.37, st, outs=(t2,))
    queue, feats, idx = f32(20, 8), f32(30, 8), rng.permutation(30)[:6].astype(np.int64)
    q2, scratch = queue.copy(), np.empty_like(queue)
    side.run("queue_push", q2, scratch, feats, idx, 20, 8, 6, st, outs=(q2,))
    xs, sc = x.copy(), (rng.random(M) < 0.5).astype(np.float32)
    side.run("scale_rows_inplace", xs, sc, M, K, st, outs=(xs,))
    R["misc"] = (t, s, t2.copy(), queue, feats, idx, q2.copy(), sc, xs.copy())
    scores = f32(96, 24, scale=0.2)
    Qkb = np.ascontiguousarray(np.exp(scores / 0.05).T)
    q_a, q_b = np.empty((96, 24), np.float32), np.empty((40, 24), np.float32)
    nbs = 1 << 20
    wss = np.empty(nbs, np.uint8)
    side.run("sinkhorn_from_q", Qkb, 0, q_a, 96, 24, 0, 96, 5, wss, nbs, st, outs=(q_a,))
    side.run("sinkhorn_from_q", np.ascontiguousarray(Qkb.T), 1, q_b, 96, 24, 16, 40, 5, wss, nbs, st, outs=(q_b,))
    R["sinkhorn_from_q"] = (scores, q_a.copy(), q_b.copy())
    v = f32(64, 40, scale=3.0)
    planes = np.empty((3, 64, 40), np.uint16)
    side.run("split_planes", v, planes, 64 * 40, 3, 64 * 40, st, outs=(planes,))
    a1, a2 = f32(777), None
    a2 = a1.copy()
    a2[[5, 99, 700]] += 1.0
    cnt = np.zeros(1, np.int64)
    side.run("count_mismatch", a1, a2, 777, cnt, st, outs=(cnt,))
    R["planes"] = (v, planes.copy(), int(cnt[0]))
    # AdamW (table of two tensors, step 3)
    import ctypes as C2
    from timetuning_amd._lib import AdamwTensor
    p0, g0, m0, v0 = f32(500), f32(500), f32(500, scale=0.1), np.abs(f32(500, scale=0.1))
    p1, g1, m1, v1 = f32(33), f32(33), f32(33, scale=0.1), np.abs(f32(33, scale=0.1))
    arrs = [a.copy() for a in (p0, g0, m0, v0, p1, g1, m1, v1)]
    if side.device is None:
        tab = (AdamwTensor * 2)(AdamwTensor(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data, 500, 1e-3, 0.1),
                                AdamwTensor(arrs[4].ctypes.data, arrs[5].ctypes.data, arrs[6].ctypes.data, arrs[7].ctypes.data, 33, 1e-2, 0.0))
        assert side.lib.tt_cpu_adamw_step(tab, 2, 3, 0.9, 0.999, 1e-8, None) == 0
        res_ad = [a.copy() for a in arrs]
    else:
        dts = [torch.from_numpy(a).cuda() for a in arrs]
        tab = (AdamwTensor * 2)(AdamwTensor(dts[0].data_ptr(), dts[1].data_ptr(), dts[2].data_ptr(), dts[3].data_ptr(), 500, 1e-3, 0.1),
                                AdamwTensor(dts[4].data_ptr(), dts[5].data_ptr(), dts[6].data_ptr(), dts[7].data_ptr(), 33, 1e-2, 0.0))
        assert side.lib.tt_adamw_step(tab, 2, 3, 0.9, 0.999, 1e-8, torch.cuda.current_stream().cuda_stream) == 0
        res_ad = [d.cpu().numpy() for d in dts]
    R["adamw"] = ((p0, g0, m0, v0, p1, g1, m1, v1), res_ad)
    return R


def _re(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(np.asarray(b, np.float64)).max(), 1e-30))


def test_round2_twins_against_torch(twin):
    """The new C twins against torch fp64 / NumPy restatements of the reference ops they cite."""
    R = _round2_cases(_Side(twin, "tt_cpu_"), 1e-6)
    x, w, b, res, y, pre = R["linear_fwd"]
    ref_pre = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert _re(pre, ref_pre) < 1e-6 and _re(y, F.gelu(torch.from_numpy(ref_pre)).numpy() + res) < 1e-6
    dy, gp, dx, dw, db = R["linear_bwd"]
    gpt = torch.from_numpy(gp).double().requires_grad_(True)
    F.gelu(gpt).sum().backward()
    assert _re(dx, (dy.astype(np.float64) @ w.astype(np.float64)) * gpt.grad.numpy()) < 1e-6
    assert _re(dw, dy.astype(np.float64).T @ x.astype(np.float64)) < 1e-6 and _re(db, dy.astype(np.float64).sum(0)) < 1e-6
    xl, g, be, yl, mu, rs, yd = R["layernorm"]
    ref = F.layer_norm(torch.from_numpy(xl).double(), (xl.shape[-1],), torch.from_numpy(g).double(), torch.from_numpy(be).double(), 1e-6).numpy()
    assert _re(yl, ref.reshape(yl.shape)) < 1e-6 and _re(yd, ref[:, 1:].reshape(yd.shape)) < 1e-6
    assert _re(mu, xl.astype(np.float64).mean(-1).reshape(-1)) < 1e-6
    xn, inv, wn = R["l2norm"]
    assert _re(xn, F.normalize(torch.from_numpy(x).double(), dim=1).numpy()) < 1e-6
    assert _re(wn, F.normalize(torch.from_numpy(w).double(), dim=1).numpy()) < 1e-6
    qkv, out, lse, probs = R["attention"]
    Fa, Na, D3 = qkv.shape
    q, k, v = torch.from_numpy(qkv).double().view(Fa, Na, 3, 2, 64).permute(2, 0, 3, 1, 4)
    s_ = q @ k.transpose(-1, -2) * 0.125
    p = torch.softmax(s_, -1)
    assert _re(probs, p.numpy()) < 1e-6 and _re(lse, torch.logsumexp(s_, -1).numpy()) < 1e-6
    assert _re(out, (p @ v).permute(0, 2, 1, 3).reshape(Fa, Na, 128).numpy()) < 1e-6
    t, s, t2, queue, feats, idx, q2, sc, xs = R["misc"]
    assert _re(t2, t * np.float32(1 - 0.37) + s * np.float32(0.37)) < 1e-6
    assert (q2[:6] == feats[idx]).all() and (q2[6:] == queue[:-6]).all() and (xs == x * sc[:, None]).all()
    scores, q_a, q_b = R["sinkhorn_from_q"]
    want = O.sinkhorn(torch.exp(torch.from_numpy(scores) / 0.05).t(), 5).numpy()
    assert _re(q_a, want) < 1e-4 and _re(q_b, want[16:56]) < 1e-4
    v_, planes, cnt = R["planes"]
    bf = torch.from_numpy(planes.astype(np.int16)).view(torch.bfloat16).double().numpy()
    assert (bf.sum(0) == v_.astype(np.float64)).all() and cnt == 3
    assert torch.equal(torch.from_numpy(planes[0].astype(np.int16)).view(torch.bfloat16), torch.from_numpy(v_).to(torch.bfloat16))
    (p0, g0, m0, v0, p1, g1, m1, v1), got = R["adamw"]
    for (p, g, m, v__, lr, wd), (gp_, gm_, gv_) in (((p0, g0, m0, v0, 1e-3, 0.1), (got[0], got[2], got[3])), ((p1, g1, m1, v1, 1e-2, 0.0), (got[4], got[6], got[7]))):
        tp = torch.nn.Parameter(torch.from_numpy(p.copy()).double())
        opt = torch.optim.AdamW([tp], lr=lr, weight_decay=wd)
        tp.grad = torch.from_numpy(g).double()
        opt.state[tp] = dict(step=torch.tensor(2.0), exp_avg=torch.from_numpy(m.copy()).double(), exp_avg_sq=torch.from_numpy(v__.copy()).double())
        opt.step()
        assert _re(gp_, tp.detach().numpy()) < 1e-6 and _re(gm_, opt.state[tp]["exp_avg"].numpy()) < 1e-6


@pytest.mark.gpu
def test_hip_library_equals_its_cpu_twin_round2(twin):
    """The HIP entry points of the hot path's row ops and matrix products against their plain-C twins, one call site."""
    from timetuning_amd import _lib

    A = _round2_cases(_Side(_lib.load(), "tt_", device="cuda"), 2e-5)
    B = _round2_cases(_Side(twin, "tt_cpu_"), 2e-5)
    for key in ("linear_fwd", "linear_bwd", "layernorm", "l2norm", "attention", "sinkhorn_from_q"):
        for i, (a, b) in enumerate(zip(A[key], B[key])):
            if isinstance(a, np.ndarray) and a.dtype == np.float32:
                assert _re(a, b) < 2e-5, (key, i, _re(a, b))
    for i, (a, b) in enumerate(zip(A["misc"], B["misc"])):
        if isinstance(a, np.ndarray):
            assert _re(a, b) < 1e-6, ("misc", i)
    assert (A["planes"][1] == B["planes"][1]).all() and A["planes"][2] == B["planes"][2] == 3     # bf16 planes bit for bit
    for a, b in zip(A["adamw"][1], B["adamw"][1]):
        assert _re(a, b) < 1e-6


# ---- second batch: backward of the row ops, attention backward, patch embedding, plane ops --------------------------------------

def _round2b_cases(side):
    st = "STREAM" if side.device else None
    rng = np.random.default_rng(11)
    f32 = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)
    R = {}
    # colsum / add
    a = f32(50, 24)
    cs = np.empty(24, np.float32)
    ws = np.empty(1 << 20, np.uint8)
    side.run("colsum", a, cs, 50, 24, ws, 1 << 20, st, outs=(cs,))
    b_ = f32(50 * 24)
    a2 = a.reshape(-1).copy()
    side.run("add_inplace", a2, b_, 50 * 24, st, outs=(a2,))
    R["colsum_add"] = (a, cs.copy(), b_, a2.copy())
    # LayerNorm backward (plain, accumulating, and with the dropped cls rows) + l2norm backward
    Fr, Nt, D = 2, 5, 40
    x, g = f32(Fr, Nt, D), 1 + 0.1 * f32(D)
    mu = x.astype(np.float64).mean(-1).reshape(-1).astype(np.float32)
    rs = (1.0 / np.sqrt(x.astype(np.float64).var(-1) + 1e-6)).reshape(-1).astype(np.float32)
    dy = f32(Fr * Nt, D)
    dx, dg, db = np.empty((Fr * Nt, D), np.float32), np.empty(D, np.float32), np.empty(D, np.float32)
    # amax_out: a zeroed slot of 16 ways x 64 floats (csrc/common.hpp amax_publish spreads its atomics; the twin uses way 0) that the kernel
    # raises to max |dx| - the maximum the next Linear's pair split scales by
    am_ln = np.zeros(16 * 64, np.float32)
    side.run("layernorm_bwd", dy, x, g, mu, rs, dx, dg, db, Fr * Nt, D, 0, 0, ws, 1 << 20, am_ln, st, outs=(dx, dg, db, am_ln))
    assert am_ln.max() == np.abs(dx).max()
    # dropped-cls form: statistics / dy only for the patch rows, dx has x's layout and its cls rows are not written
    keep = np.array([r for r in range(Fr * Nt) if r % Nt != 0])
    dyd = np.ascontiguousarray(dy[keep])
    dxd = np.zeros((Fr * Nt, D), np.float32)
    side.run("layernorm_bwd", dyd, x, g, np.ascontiguousarray(mu[keep]), np.ascontiguousarray(rs[keep]), dxd, None, None, Fr * (Nt - 1), D, 1, Nt, ws, 1 << 20,
             None, st, outs=(dxd,))
    R["ln_bwd"] = (x, g, dy, dx.copy(), dg.copy(), db.copy(), keep, dxd.copy(), am_ln.max(keepdims=True))
    xn = x.reshape(-1, D) / np.linalg.norm(x.reshape(-1, D), axis=1, keepdims=True)
    inv = (1.0 / np.linalg.norm(x.reshape(-1, D), axis=1)).astype(np.float32)
    dxn, dxl = f32(Fr * Nt, D), np.empty((Fr * Nt, D), np.float32)
    am_l2 = np.zeros(16 * 64, np.float32)
    side.run("l2norm_bwd", dxn, xn.astype(np.float32), inv, dxl, Fr * Nt, D, am_l2, st, outs=(dxl, am_l2))
    assert am_l2.max() == np.abs(dxl).max()
    R["l2_bwd"] = (x.reshape(-1, D), dxn, dxl.copy(), am_l2.max(keepdims=True))
    # attention backward from the forward's saved out / lse
    Fa, Na, H = 2, 21, 2
    qkv = f32(Fa, Na, 3 * H * 64, scale=0.5)
    out, lse = np.empty((Fa, Na, H * 64), np.float32), np.empty((Fa, H, Na), np.float32)
    side.run("attention_fwd", qkv, out, lse, None, Fa, Na, H, 64, 0.125, st, outs=(out, lse))
    dout, dqkv = f32(Fa, Na, H * 64), np.empty_like(qkv)
    nbw = 1 << 24
    wsa = np.empty(nbw, np.uint8)
    am_at = np.zeros(16 * 64, np.float32)
    side.run("attention_bwd", qkv, out, dout, lse, dqkv, Fa, Na, H, 64, 0.125, wsa, nbw, am_at, st, outs=(dqkv, am_at))
    assert am_at.max() == np.abs(dqkv).max()
    R["att_bwd"] = (qkv, dout, dqkv.copy(), am_at.max(keepdims=True))
    dqkv_p, am_p, flag_p = np.empty_like(qkv), np.zeros(16 * 64, np.float32), np.zeros(1, np.int32)
    side.run("attention_bwd_pairs", qkv, out, dout, lse, dqkv_p, Fa, Na, H, 64, 0.125, None, wsa, nbw, flag_p, am_p, st, outs=(dqkv_p, am_p))
    R["att_bwd_pairs"] = (qkv, dout, dqkv_p.copy())
    dqkv_b = np.empty_like(qkv)
    side.run("attention_bwd_bf16", qkv, out, dout, lse, dqkv_b, Fa, Na, H, 64, 0.125, wsa, nbw, st, outs=(dqkv_b,))
    R["att_bwd_bf16"] = (qkv, dout, dqkv_b.copy())
    # patch embedding with a frame map
    Fp, C, Hh, Ww, P, Dm = 3, 3, 32, 48, 16, 24
    img, w, b, cls, pos = f32(4, C, Hh, Ww), f32(Dm, C * P * P, scale=0.05), f32(Dm), f32(Dm), f32(1 + 6, Dm)
    fmap = np.array([2, 0, 3], np.int32)
    tok = np.empty((Fp, 7, Dm), np.float32)
    side.run("patch_embed_fwd", img, fmap, w, b, cls, pos, tok, Fp, C, Hh, Ww, P, Dm, st, outs=(tok,))
    R["patch"] = (img, fmap, w, b, cls, pos, tok.copy())
    xa, sc, sh = f32(30, 7), f32(7), f32(7)
    xa2 = xa.copy()
    side.run("affine_cols_inplace", xa2, sc, sh, 30, 7, st, outs=(xa2,))
    R["affine"] = (xa, sc, sh, xa2.copy())
    # plane ops
    xt = f32(70, 24)
    tp = np.empty((24, 128), np.uint16)
    side.run("transpose_planes", xt, tp, 70, 24, 128, st, outs=(tp,))
    xl, gl, bl = f32(3, 5, 64), 1 + 0.1 * f32(64), 0.1 * f32(64)
    lp = np.empty((3, 15, 64), np.uint16)
    side.run("layernorm_fwd_planes", xl, gl, bl, lp, 15 * 64, 3, None, None, 15, 64, 1e-6, 0, st, outs=(lp,))
    tp2, cs2 = np.empty((24, 128), np.uint16), np.empty(24, np.float32)
    nbt = max(int(getattr(side.lib, side.prefix + "transpose_planes_colsum_workspace_bytes")(70, 24, 128)), 16)
    wst = np.empty(nbt, np.uint8)
    side.run("transpose_planes_colsum", xt, tp2, 70, 24, 128, cs2, wst, nbt, st, outs=(tp2, cs2))
    R["transpose_colsum"] = (xt, tp2.copy(), cs2.copy())
    R["planes_misc"] = (xt, tp.copy(), xl, gl, bl, lp.copy())
    M, N, K = 70, 64, 128
    xs_, w_, bb, rr = f32(M, K), f32(N, K, scale=0.1), f32(N), f32(M, N)
    res = {}
    for P_ in (1, 3):
        xp, wp = np.empty((P_, M, K), np.uint16), np.empty((P_, N, K), np.uint16)
        side.run("split_planes", xs_, xp, M * K, P_, M * K, st, outs=(xp,))
        side.run("split_planes", w_, wp, N * K, P_, N * K, st, outs=(wp,))
        y, pre, yp = np.empty((M, N), np.float32), np.empty((M, N), np.float32), np.empty((P_, M, N), np.uint16)
        side.run("linear_fwd_planes", xp, M * K, wp, N * K, P_, bb, rr, y, pre, yp, M * N, P_, M, N, K, 1, None, 0, st, outs=(y, pre, yp))
        res[P_] = (xp.copy(), wp.copy(), y.copy(), pre.copy(), yp.copy())
    R["plane_gemm"] = (xs_, w_, bb, rr, res)
    qb = np.empty((2, 21, 3 * 128), np.uint16)
    side.run("split_planes", f32(2, 21, 3 * 128, scale=0.6), qb, 2 * 21 * 384, 1, 2 * 21 * 384, st, outs=(qb,))
    ob = np.empty((2, 21, 128), np.uint16)
    side.run("attention_fwd_bf16", qb, ob, 2, 21, 2, 64, 0.125, st, outs=(ob,))
    R["att_bf16"] = (qb, ob.copy())
    # patch embedding on bf16 operands (frame map, 2 x 3 grid, D = 64)
    Fp, C, Hh, Ww, P, Dm = 3, 3, 32, 48, 16, 64
    img, w, b, cls, pos = f32(4, C, Hh, Ww), f32(Dm, C * P * P, scale=0.05), f32(Dm), f32(Dm), f32(1 + 6, Dm)
    fmap = np.array([2, 0, 3], np.int32)
    wpl = np.empty((1, Dm, C * P * P), np.uint16)
    side.run("split_planes", w, wpl, Dm * C * P * P, 1, Dm * C * P * P, st, outs=(wpl,))
    tok = np.empty((Fp, 7, Dm), np.float32)
    nbp = max(int(getattr(side.lib, side.prefix + "patch_embed_planes_workspace_bytes")(Fp, C, Hh, Ww, P)), 16)
    wsp = np.empty(nbp, np.uint8)
    side.run("patch_embed_fwd_planes", img, fmap, wpl, b, cls, pos, tok, Fp, C, Hh, Ww, P, Dm, wsp, nbp, st, outs=(tok,))
    R["patch_bf16"] = (img, fmap, w, wpl.copy(), b, cls, pos, tok.copy())
    # ... and on fp16-pair operands
    wpr = np.empty((Dm, 2 * C * P * P), np.uint16)
    side.run("split_pairs", w, wpr, Dm * C * P * P, None, st, outs=(wpr,))
    tokp = np.empty((Fp, 7, Dm), np.float32)
    nbq = max(int(getattr(side.lib, side.prefix + "patch_embed_pairs_workspace_bytes")(Fp, C, Hh, Ww, P)), 16)
    wsq = np.empty(nbq, np.uint8)
    side.run("patch_embed_fwd_pairs", img, fmap, wpr, b, cls, pos, tokp, Fp, C, Hh, Ww, P, Dm, wsq, nbq, None, st, outs=(tokp,))
    R["patch_pairs"] = (tok.copy(), tokp.copy())
    return R


def _bf(a):
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16).double().numpy()


def test_round2b_twins_against_torch(twin):
    R = _round2b_cases(_Side(twin, "tt_cpu_"))
    a, cs, b_, a2 = R["colsum_add"]
    assert _re(cs, a.astype(np.float64).sum(0)) < 1e-6 and np.array_equal(a2, a.reshape(-1) + b_)
    x, g, dy, dx, dg, db, keep, dxd, _ = R["ln_bwd"]
    xt = torch.from_numpy(x).double().requires_grad_(True)
    gt = torch.from_numpy(g).double().requires_grad_(True)
    bt = torch.zeros(x.shape[-1], dtype=torch.float64, requires_grad=True)
    (F.layer_norm(xt, (x.shape[-1],), gt, bt, 1e-6).reshape(-1, x.shape[-1]) * torch.from_numpy(dy).double()).sum().backward()
    assert _re(dx, xt.grad.reshape(dx.shape).numpy()) < 1e-5 and _re(dg, gt.grad.numpy()) < 1e-5 and _re(db, bt.grad.numpy()) < 1e-6
    xt.grad = None
    y2 = F.layer_norm(xt, (x.shape[-1],), gt.detach(), bt.detach(), 1e-6).reshape(-1, x.shape[-1])[torch.from_numpy(keep)]
    (y2 * torch.from_numpy(dy[keep]).double()).sum().backward()
    assert _re(dxd, xt.grad.reshape(dxd.shape).numpy()) < 1e-5 and (dxd[::5] == 0).all()
    xx, dxn, dxl, _ = R["l2_bwd"]
    xt = torch.from_numpy(xx).double().requires_grad_(True)
    (F.normalize(xt, dim=1) * torch.from_numpy(dxn).double()).sum().backward()
    assert _re(dxl, xt.grad.numpy()) < 1e-5
    qkv, dout, dqkv, _ = R["att_bwd"]
    qt = torch.from_numpy(qkv).double().requires_grad_(True)
    q, k, v = qt.view(2, 21, 3, 2, 64).permute(2, 0, 3, 1, 4)
    o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(2, 21, 128)
    (o * torch.from_numpy(dout).double()).sum().backward()
    assert _re(dqkv, qt.grad.numpy()) < 1e-5
    assert _re(R["att_bwd_pairs"][2], qt.grad.numpy()) < 1e-5   # fp16 pairs: fp32-class
    # bf16 products: near the exact gradient (the rounding of q, k, v, dout, P and dS), and NOT equal to it
    e_bf = _re(R["att_bwd_bf16"][2], qt.grad.numpy())
    assert 1e-4 < e_bf < 2e-2, e_bf
    img, fmap, w, b, cls, pos, tok = R["patch"]
    conv = F.conv2d(torch.from_numpy(img[fmap]).double(), torch.from_numpy(w).double().view(24, 3, 16, 16), torch.from_numpy(b).double(), stride=16)
    ref = torch.cat([torch.from_numpy(cls).double().expand(3, 1, 24), conv.flatten(2).transpose(1, 2)], 1) + torch.from_numpy(pos).double()
    assert _re(tok, ref.numpy()) < 1e-6
    xa, sc, sh, xa2 = R["affine"]
    assert np.array_equal(xa2, xa * sc + sh)
    xt_, tp, xl, gl, bl, lp = R["planes_misc"]
    assert np.array_equal(_bf(tp)[:, :70], torch.from_numpy(xt_).to(torch.bfloat16).double().numpy().T) and (tp[:, 70:] == 0).all()
    xt2, tp2, cs2 = R["transpose_colsum"]
    assert np.array_equal(tp2, tp) and _re(cs2, xt2.astype(np.float64).sum(0)) < 1e-6
    lnref = F.layer_norm(torch.from_numpy(xl).double(), (64,), torch.from_numpy(gl).double(), torch.from_numpy(bl).double(), 1e-6).numpy()
    assert _re(_bf(lp).sum(0), lnref.reshape(15, 64)) < 1e-6
    xs_, w_, bb, rr, res = R["plane_gemm"]
    for P_, (xp, wp, y, pre, yp) in res.items():
        ref_pre = sum(_bf(xp[i]) @ _bf(wp[j]).T for i in range(P_) for j in range(P_ - i)) + bb
        assert _re(pre, ref_pre) < 1e-6 and _re(y, F.gelu(torch.from_numpy(ref_pre)).numpy() + rr) < 1e-6
        assert _re(_bf(yp).sum(0), y) < (1e-7 if P_ == 3 else 2.0 ** -8)
    assert _re(res[3][3], xs_.astype(np.float64) @ w_.astype(np.float64).T + bb) < 2e-6          # three planes: fp32-accurate
    qb, ob = R["att_bf16"]
    qf = torch.from_numpy(_bf(qb))
    q, k, v = qf.view(2, 21, 3, 2, 64).permute(2, 0, 3, 1, 4)
    o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(2, 21, 128)
    assert _re(_bf(ob), o.numpy()) < 1.5e-2
    img, fmap, w, wpl, b, cls, pos, tok = R["patch_bf16"]
    assert np.array_equal(_bf(wpl)[0], torch.from_numpy(w).to(torch.bfloat16).double().numpy())
    conv = F.conv2d(torch.from_numpy(img[fmap]).to(torch.bfloat16).double(), torch.from_numpy(_bf(wpl)[0]).view(64, 3, 16, 16),
                    torch.from_numpy(b).double(), stride=16)     # the conv of the ROUNDED operands, exactly
    ref = torch.cat([torch.from_numpy(cls).double().expand(3, 1, 64), conv.flatten(2).transpose(1, 2)], 1) + torch.from_numpy(pos).double()
    assert _re(tok, ref.numpy()) < 1e-6
    full = F.conv2d(torch.from_numpy(img[fmap]).double(), torch.from_numpy(w).double().view(64, 3, 16, 16), torch.from_numpy(b).double(), stride=16)
    assert _re(tok[:, 1:], (full.flatten(2).transpose(1, 2) + torch.from_numpy(pos).double()[1:]).numpy()) < 1e-2     # bf16 operands
    tokp = R["patch_pairs"][1]                                                                                          # pair operands: fp32-class
    assert _re(tokp[:, 1:], (full.flatten(2).transpose(1, 2) + torch.from_numpy(pos).double()[1:]).numpy()) < 1e-6
    assert np.abs(tokp[:, 0] - (cls + pos[0])).max() < 1e-6


@pytest.mark.gpu
def test_hip_library_equals_its_cpu_twin_round2b(twin):
    from timetuning_amd import _lib

    A = _round2b_cases(_Side(_lib.load(), "tt_", device="cuda"))
    B = _round2b_cases(_Side(twin, "tt_cpu_"))
    for key in ("colsum_add", "ln_bwd", "l2_bwd", "att_bwd", "att_bwd_pairs", "patch", "affine"):
        for i, (a, b) in enumerate(zip(A[key], B[key])):
            if isinstance(a, np.ndarray) and a.dtype == np.float32:
                assert _re(a, b) < 2e-5, (key, i, _re(a, b))
    assert _re(A["att_bwd_bf16"][2], B["att_bwd_bf16"][2]) < 3e-3          # same rounding points; P / dS may round one bf16 ulp apart
    assert np.array_equal(A["planes_misc"][1], B["planes_misc"][1])                         # transposed bf16 image: bit for bit
    assert np.array_equal(A["transpose_colsum"][1], B["transpose_colsum"][1]) and _re(A["transpose_colsum"][2], B["transpose_colsum"][2]) < 2e-6
    assert _re(_bf(A["planes_misc"][5]).sum(0), _bf(B["planes_misc"][5]).sum(0)) < 2e-6       # LayerNorm planes: same value to fp32 rounding
    for P_ in (1, 3):
        for i in (2, 3):
            assert _re(A["plane_gemm"][4][P_][i], B["plane_gemm"][4][P_][i]) < 2e-5, (P_, i)
    assert _re(_bf(A["att_bf16"][1]), _bf(B["att_bf16"][1])) < 1.5e-2
    assert np.array_equal(A["patch_bf16"][3], B["patch_bf16"][3]) and _re(A["patch_bf16"][7], B["patch_bf16"][7]) < 2e-5
    assert _re(A["patch_pairs"][1], B["patch_pairs"][1]) < 2e-6


# ---- third batch: generic GEMM, position-table resampling, evaluator resampling, k-means sums, plane backward products, the
# ---- foreground mask and label propagation ----------------------------------------------------------------------------------

def _round2c_cases(side):
    st = "STREAM" if side.device else None
    rng = np.random.default_rng(23)
    f32 = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)
    R = {}
    # generic GEMM: k-major A, n-major B, alpha, batch of 2
    M, N, K = 37, 29, 24
    A, B = f32(2, K, 40), f32(2, K, 32)
    Cm = np.zeros((2, M, 36), np.float32)
    side.run("gemm_f32", A, B, Cm, M, N, K, 40, 32, 36, 1, 1, 0.5, 2, K * 40, K * 32, M * 36, st, outs=(Cm,))
    R["gemm"] = (A, B, Cm.copy(), (M, N, K))
    # position table 5x5 -> 3x7 tokens (dino_vision_transformer.py:219-231)
    g, gh, gw, D = 5, 3, 7, 8
    pos, out = f32(1 + g * g, D), np.empty((1 + gh * gw, D), np.float32)
    sh, sw = (gh + 0.1) / g, (gw + 0.1) / g
    side.run("pos_embed_interpolate", pos, out, g, gh, gw, D, sh, sw, st, outs=(out,))
    R["pos"] = (pos, out.copy(), (g, gh, gw, D, sh, sw))
    # evaluator resampling
    Mm, gg, C, Rr, Kk = 2, 6, 5, 20, 7
    x, up = f32(Mm, gg * gg, C), np.empty((Mm, Rr * Rr, C), np.float32)
    side.run("upsample_bilinear_tokens", x, up, Mm, gg, C, Rr, st, outs=(up,))
    maps, lab = f32(Mm, gg * gg, Kk), np.empty((Mm, Rr, Rr), np.int64)
    side.run("upsample_argmax_f32", maps, lab, Mm, gg, Kk, Rr, st, outs=(lab,))
    R["upsample"] = (x, up.copy(), maps, lab.copy(), (Mm, gg, C, Rr, Kk))
    # k-means sums
    P, d, k = 5000, 6, 9
    pts, lbl = f32(P, d), rng.integers(0, k, P).astype(np.int32)
    sums, counts = np.empty((k, d), np.float64), np.empty(k, np.int64)
    nb = 1 << 24
    ws = np.empty(nb, np.uint8)
    side.run("kmeans_accumulate", pts, lbl, sums, counts, P, d, k, ws, nb, st, outs=(sums, counts))
    R["kmeans"] = (pts, lbl, sums.copy(), counts.copy())
    # plane backward products (one plane: the bf16 path)
    M2, N2, K2, Mpad = 100, 64, 128, 128
    dy, w, xx, pre = f32(M2, N2), f32(N2, K2, scale=0.1), f32(M2, K2), f32(M2, K2)
    dyp, wT = np.empty((1, M2, N2), np.uint16), np.empty((K2, N2), np.uint16)
    side.run("split_planes", dy, dyp, M2 * N2, 1, M2 * N2, st, outs=(dyp,))
    side.run("transpose_planes", w, wT, N2, K2, N2, st, outs=(wT,))
    dx = np.empty((M2, K2), np.float32)
    side.run("linear_bwd_data_planes", dyp, M2 * N2, wT, K2 * N2, 1, pre, dx, M2, N2, K2, None, 0, st, outs=(dx,))
    dyT, xT = np.empty((N2, Mpad), np.uint16), np.empty((K2, Mpad), np.uint16)
    side.run("transpose_planes", dy, dyT, M2, N2, Mpad, st, outs=(dyT,))
    side.run("transpose_planes", xx, xT, M2, K2, Mpad, st, outs=(xT,))
    dw = np.empty((N2, K2), np.float32)
    side.run("linear_bwd_weight_planes", dyT, N2 * Mpad, xT, K2 * Mpad, 1, dw, N2, K2, Mpad, ws, nb, st, outs=(dw,))
    R["plane_bwd"] = (dy, w, xx, pre, dx.copy(), dw.copy())
    # foreground mask: from qkv and from the probabilities
    Fm, gm, H, hd = 3, 14, 2, 64
    Nm = gm * gm + 1
    qkv = f32(Fm, Nm, 3 * H * hd, scale=0.6)
    mask, blur, margin = (np.empty((Fm, gm * gm), np.float32) for _ in range(3))
    side.run("foreground_mask", qkv, mask, blur, margin, Fm, Nm, H, hd, gm, 0.125, 0.65, 0.6, 7, st, outs=(mask, blur, margin))
    q, kx = qkv.reshape(Fm, Nm, 3, H, hd)[:, 0, 0], qkv.reshape(Fm, Nm, 3, H, hd)[:, :, 1]
    logits = np.einsum("fhd,fjhd->fhj", q.astype(np.float64), kx.astype(np.float64)) * 0.125
    probs = np.exp(logits - logits.max(-1, keepdims=True))
    probs = np.ascontiguousarray((probs / probs.sum(-1, keepdims=True)).astype(np.float32))   # (einsum's output order is not C)
    mask2, blur2, margin2 = (np.empty((Fm, gm * gm), np.float32) for _ in range(3))
    side.run("foreground_mask_from_probs", probs, mask2, blur2, margin2, Fm, Nm, H, gm, 0.65, 0.6, 7, st, outs=(mask2, blur2, margin2))
    R["mask"] = (probs, mask.copy(), blur.copy(), margin.copy(), mask2.copy(), blur2.copy(), margin2.copy(), (Fm, gm, H))
    # label propagation: 5 frames, 2 context frames (queue filling, then full), hard labels + last map + all maps
    bs, fs, gl, Dl, Kl, nlast = 2, 5, 6, 16, 4, 2
    nl = gl * gl
    feats = f32(fs, bs, nl, Dl)
    for t in range(1, fs):
        feats[t] = 0.8 * feats[t - 1] + 0.2 * feats[t]
    xn = (feats / np.linalg.norm(feats, axis=-1, keepdims=True)).astype(np.float32)
    seg0 = np.exp(f32(bs, nl, Kl) * 2)
    seg0 = (seg0 / seg0.sum(-1, keepdims=True)).astype(np.float32)
    labels, last, allm = np.empty((bs, nl), np.int64), np.empty((bs, nl, Kl), np.float64), np.empty((fs - 1, bs, nl, Kl), np.float64)
    side.run("label_propagate", xn, seg0, labels, last, bs, fs, gl, Dl, Kl, nlast, 2, 3, 0.1, 0, ws, nb, st, outs=(labels, last))
    side.run("label_propagate_maps", xn, seg0, allm, bs, fs, gl, Dl, Kl, nlast, 2, 3, 0.1, 0, ws, nb, st, outs=(allm,))
    R["lp"] = (xn, seg0, labels.copy(), last.copy(), allm.copy(), (bs, fs, gl, Dl, Kl, nlast))
    return R


def test_round2c_twins_against_references(twin):
    R = _round2c_cases(_Side(twin, "tt_cpu_"))
    A, B, Cm, (M, N, K) = R["gemm"]
    for z in range(2):
        ref = 0.5 * A[z, :, :M].T.astype(np.float64) @ B[z, :, :N].astype(np.float64)
        assert _re(Cm[z, :, :N], ref) < 1e-6 and (Cm[z, :, N:] == 0).all()
    pos, out, (g, gh, gw, D, sh, sw) = R["pos"]
    grid = torch.from_numpy(pos[1:]).view(1, g, g, D).permute(0, 3, 1, 2)
    ref = F.interpolate(grid, scale_factor=(sh, sw), mode="bicubic").permute(0, 2, 3, 1).reshape(-1, D)   # dino_vision_transformer.py:224-228
    assert ref.shape[0] == gh * gw and np.array_equal(out[0], pos[0]) and _re(out[1:], ref.numpy()) < 1e-5
    x, up, maps, lab, (Mm, gg, C, Rr, Kk) = R["upsample"]
    ref = F.interpolate(torch.from_numpy(x).double().view(Mm, gg, gg, C).permute(0, 3, 1, 2), (Rr, Rr), mode="bilinear").float()
    assert _re(up, ref.permute(0, 2, 3, 1).reshape(Mm, Rr * Rr, C).numpy()) < 1e-7
    ref = F.interpolate(torch.from_numpy(maps).view(Mm, gg, gg, Kk).permute(0, 3, 1, 2), (Rr, Rr), mode="bilinear").argmax(1)
    assert (lab != ref.numpy()).mean() < 2e-3
    pts, lbl, sums, counts = R["kmeans"]
    for j in range(len(counts)):
        assert counts[j] == (lbl == j).sum() and _re(sums[j], pts[lbl == j].astype(np.float64).sum(0)) < 1e-12
    dy, w, xx, pre, dx, dw = R["plane_bwd"]
    b = lambda a: torch.from_numpy(a).to(torch.bfloat16).double().numpy()
    pd = torch.from_numpy(pre).double()
    gelu_grad = (0.5 * (1 + torch.erf(pd / 2 ** 0.5)) + pd * torch.exp(-0.5 * pd * pd) / (2 * np.pi) ** 0.5).numpy()
    assert _re(dx, (b(dy) @ b(w)) * gelu_grad) < 1e-5 and _re(dw, b(dy).T @ b(xx)) < 1e-6
    probs, mask, blur, margin, mask2, blur2, margin2, (Fm, gm, H) = R["mask"]
    full = torch.zeros(Fm, H, gm * gm + 1, gm * gm + 1)
    full[:, :, 0, :] = torch.from_numpy(probs)
    omask, oblur, omargin = O.process_attentions(full, gm, 0.65, 0.6, return_blurred=True)
    for mk, bl in ((mask, blur), (mask2, blur2)):
        assert _re(bl, oblur.numpy()) < 1e-5
        diff = mk != omask.reshape(Fm, -1).numpy()
        assert not (diff & (omargin.numpy() > 2e-6)).any() and diff.mean() < 0.01
    assert 0.2 < mask.mean() < 0.9
    xn, seg0, labels, last, allm, (bs, fs, gl, Dl, Kl, nlast) = R["lp"]
    assert np.array_equal(last, allm[-1])
    for bi in range(bs):
        seed = torch.from_numpy(seg0[bi]).view(gl, gl, Kl).permute(2, 0, 1).unsqueeze(0)
        ref = torch.stack(O.propagate_labels(nlast, 2, 3, gl, torch.from_numpy(xn[:, bi]), seed)).reshape(fs - 1, Kl, gl * gl).transpose(1, 2).numpy()
        bad = np.abs(allm[:, bi] - ref).max(-1) > 1e-5 * np.abs(ref).max()
        assert bad.mean() <= 0.01
        assert (labels[bi] != ref[-1].argmax(1)).mean() <= 0.03
    # the table-driven scale: g *= scale for every entry
    from timetuning_amd._lib import AdamwTensor
    gs = [np.arange(5, dtype=np.float32), np.ones(3, np.float32)]
    tab = (AdamwTensor * 2)(*[AdamwTensor(None, ptr(a), None, None, a.size, 0.0, 0.0) for a in gs])
    sc = np.array([0.25], np.float32)
    assert twin.tt_cpu_scale_tensors(tab, 2, ptr(sc), None) == 0
    assert np.array_equal(gs[0], np.arange(5) * 0.25) and np.array_equal(gs[1], np.full(3, 0.25, np.float32))


@pytest.mark.gpu
def test_hip_library_equals_its_cpu_twin_round2c(twin):
    from timetuning_amd import _lib

    A = _round2c_cases(_Side(_lib.load(), "tt_", device="cuda"))
    B = _round2c_cases(_Side(twin, "tt_cpu_"))
    assert _re(A["gemm"][2], B["gemm"][2]) < 2e-5 and _re(A["pos"][1], B["pos"][1]) < 1e-5
    assert _re(A["upsample"][1], B["upsample"][1]) < 1e-6 and (A["upsample"][3] != B["upsample"][3]).mean() < 2e-3
    # (the kernel sums a block's points in fp32 in LDS - as faiss accumulates its centroids in float - and folds the blocks in fp64)
    assert np.array_equal(A["kmeans"][3], B["kmeans"][3]) and _re(A["kmeans"][2], B["kmeans"][2]) < 1e-6
    assert _re(A["plane_bwd"][4], B["plane_bwd"][4]) < 2e-5 and _re(A["plane_bwd"][5], B["plane_bwd"][5]) < 2e-5
    for i in (1, 4):                                         # masks: equal except within the cut's rounding margin
        diff = A["mask"][i] != B["mask"][i]
        assert not (diff & (B["mask"][i + 2] > 2e-6)).any() and diff.mean() < 0.01
        assert _re(A["mask"][i + 1], B["mask"][i + 1]) < 1e-5
    bad = np.abs(A["lp"][4] - B["lp"][4]).max(-1) > 1e-5 * np.abs(B["lp"][4]).max()
    assert bad.mean() <= 0.01 and (A["lp"][2] != B["lp"][2]).mean() <= 0.03 and np.array_equal(A["lp"][3], A["lp"][4][-1])


def test_round4_pair_twins_against_numpy(twin):
    """The fp16-pair twins (the "f16x3" mode's operand format and products) against NumPy / torch restatements on the CPU."""
    rs = np.random.RandomState(4)
    M, N, K = 37, 64, 96
    x, w, b = rs.randn(M, K).astype(np.float32), (0.1 * rs.randn(N, K)).astype(np.float32), (0.1 * rs.randn(N)).astype(np.float32)
    xp, wp = np.empty((M, 2 * K), np.uint16), np.empty((N, 2 * K), np.uint16)
    assert twin.tt_cpu_split_pairs(ptr(x), ptr(xp), M * K, None, None) == 0 and twin.tt_cpu_split_pairs(ptr(w), ptr(wp), N * K, None, None) == 0
    # layout and values: groups of 32 as [hi x 32][lo x 32]; hi = fp16(x) (torch's own conversion), lo = fp16((x - hi) 2^11)
    g = xp.reshape(M, K // 32, 2, 32)
    hi_ref = torch.from_numpy(x).to(torch.float16)
    assert np.array_equal(g[:, :, 0, :].reshape(M, K), hi_ref.numpy().view(np.uint16))
    lo_ref = ((torch.from_numpy(x) - hi_ref.float()) * 2048).to(torch.float16)
    assert np.array_equal(g[:, :, 1, :].reshape(M, K), lo_ref.numpy().view(np.uint16))
    back = np.empty((M, K), np.float32)
    assert twin.tt_cpu_join_pairs(ptr(xp), ptr(back), M * K, None) == 0
    assert np.abs(back - x).max() <= 2.0 ** -22 * np.abs(x).max()
    y, pre, yp = np.empty((M, N), np.float32), np.empty((M, N), np.float32), np.empty((M, 2 * N), np.uint16)
    assert twin.tt_cpu_linear_fwd_pairs(ptr(xp), ptr(wp), ptr(b), None, ptr(y), ptr(pre), ptr(yp), M, N, K, 1, None, 0, None, None) == 0
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert np.abs(pre - ref).max() / np.abs(ref).max() < 5e-7
    assert np.abs(y - F.gelu(torch.from_numpy(ref)).numpy()).max() / np.abs(ref).max() < 5e-7
    # transposes: zero padding beyond R, the same bits as splitting the transposed matrix
    t, row, sums = np.empty((K, 2 * 64), np.uint16), np.empty((M, 2 * K), np.uint16), np.empty(K, np.float32)
    assert twin.tt_cpu_split_pairs_dual(ptr(x), ptr(t), ptr(row), ptr(sums), None, M, K, 64, None, 0, None, None) == 0
    xt = np.zeros((K, 64), np.float32); xt[:, :M] = x.T
    tref = np.empty((K, 128), np.uint16)
    assert twin.tt_cpu_split_pairs(ptr(xt), ptr(tref), K * 64, None, None) == 0
    assert np.array_equal(t, tref) and np.array_equal(row, xp) and np.allclose(sums, x.sum(0), atol=1e-5)
    t2 = np.empty((K, 128), np.uint16)
    assert twin.tt_cpu_transpose_pairs(ptr(xp), ptr(t2), M, K, 64, None) == 0 and np.array_equal(t2, t)
    # backward products
    dy = (1e-3 * rs.randn(M, N)).astype(np.float32)
    dyT, dyr = np.empty((N, 128), np.uint16), np.empty((M, 2 * N), np.uint16)
    assert twin.tt_cpu_split_pairs_dual(ptr(dy), ptr(dyT), ptr(dyr), None, None, M, N, 64, None, 0, None, None) == 0
    wT = np.empty((K, 2 * N), np.uint16)
    assert twin.tt_cpu_split_pairs_dual(ptr(w), ptr(wT), None, None, None, N, K, N, None, 0, None, None) == 0
    dx, dw = np.empty((M, K), np.float32), np.empty((N, K), np.float32)
    am = np.zeros(16 * 64, np.float32)
    assert twin.tt_cpu_linear_bwd_data_pairs(ptr(dyr), ptr(wT), None, ptr(dx), None, M, N, K, None, 0, ptr(am), None) == 0
    assert am.max() == np.abs(dx).max()
    assert twin.tt_cpu_linear_bwd_weight_pairs(ptr(dyT), ptr(t), ptr(dw), None, N, K, 64, None, 0, None) == 0
    dx_ref, dw_ref = dy.astype(np.float64) @ w, dy.astype(np.float64).T @ x
    assert np.abs(dx - dx_ref).max() / np.abs(dx_ref).max() < 5e-7 and np.abs(dw - dw_ref).max() / np.abs(dw_ref).max() < 5e-7
    # the same weight gradient from ROW pairs (gemm_pairs_tn.hip's twin); the column sums without a transposed output
    dw_tn = np.empty((N, K), np.float32)
    assert twin.tt_cpu_linear_bwd_weight_pairs_tn(ptr(dyr), ptr(xp), ptr(dw_tn), None, N, K, M, None, 0, None) == 0
    assert np.abs(dw_tn - dw_ref).max() / np.abs(dw_ref).max() < 5e-7 and np.abs(dw_tn - dw).max() <= 1e-6 * np.abs(dw).max()
    assert twin.tt_cpu_linear_bwd_weight_pairs_tn_ok(128, 256, 5) == 1 and twin.tt_cpu_linear_bwd_weight_pairs_tn_ok(64, 256, 5) == 0
    import ctypes as C
    m_t, m_row = np.empty((K, 128), np.uint16), np.empty((M, 2 * K), np.uint16)
    arr = lambda *ps: (C.c_void_p * len(ps))(*[C.c_void_p(p_) if p_ is not None else None for p_ in ps])
    ints = lambda *v: (C.c_int * len(v))(*v)
    assert twin.tt_cpu_split_pairs_dual_multi(arr(x.ctypes.data, x.ctypes.data), arr(m_t.ctypes.data, None), arr(None, m_row.ctypes.data),
                                              ints(M, M), ints(K, K), ints(64, 64), 2, None, None) == 0
    assert np.array_equal(m_t, t) and np.array_equal(m_row, xp)
    dyr2, cs = np.empty((M, 2 * N), np.uint16), np.empty((N,), np.float32)
    assert twin.tt_cpu_split_pairs_dual(ptr(dy), None, ptr(dyr2), ptr(cs), None, M, N, 64, None, 0, None, None) == 0
    assert np.array_equal(dyr2, dyr) and np.allclose(cs, dy.sum(0), atol=1e-6)
    # a gradient far below fp16's normal range: the scaled split (a power of two S that brings max |dy| into [2^13, 2^14), the products
    # divided by S) keeps fp32-class accuracy where the plain split is left with fp16 subnormals
    tiny = (dy * 1e-4).astype(np.float32)                                  # |tiny| ~ 1e-7
    ref_dw, ref_dx = tiny.astype(np.float64).T @ x, tiny.astype(np.float64) @ w
    S = np.zeros(1, np.float32)
    tr, cs2 = np.empty((M, 2 * N), np.uint16), np.empty((N,), np.float32)
    assert twin.tt_cpu_split_pairs_dual(ptr(tiny), None, ptr(tr), ptr(cs2), ptr(S), M, N, 64, None, 0, None, None) == 0
    amax = np.abs(tiny).max()
    assert S[0] == 2.0 ** (13 - np.floor(np.log2(amax))) and 2 ** 13 <= amax * S[0] < 2 ** 14 and np.allclose(cs2, tiny.sum(0), rtol=1e-5, atol=0)
    dws, dxs = np.empty((N, K), np.float32), np.empty((M, K), np.float32)
    assert twin.tt_cpu_linear_bwd_weight_pairs_tn(ptr(tr), ptr(xp), ptr(dws), ptr(S), N, K, M, None, 0, None) == 0
    assert twin.tt_cpu_linear_bwd_data_pairs(ptr(tr), ptr(wT), None, ptr(dxs), ptr(S), M, N, K, None, 0, None, None) == 0
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(dws, ref_dw) < 5e-7 and rel(dxs, ref_dx) < 5e-7
    tu, dwu = np.empty((M, 2 * N), np.uint16), np.empty((N, K), np.float32)
    assert twin.tt_cpu_split_pairs_dual(ptr(tiny), None, ptr(tu), None, None, M, N, 64, None, 0, None, None) == 0
    assert twin.tt_cpu_linear_bwd_weight_pairs_tn(ptr(tu), ptr(xp), ptr(dwu), None, N, K, M, None, 0, None) == 0
    assert rel(dwu, ref_dw) > 20 * rel(dws, ref_dw)                        # (what the scale is for)
    # the column sums left as partials of 64-row blocks, folded by the weight gradient's call (round 5: one fold launch per Linear)
    parts, tr3, S3 = np.full((1, N), np.nan, np.float32), np.empty((M, 2 * N), np.uint16), np.zeros(1, np.float32)
    amx = np.zeros(16 * 64, np.float32); amx[0] = np.abs(tiny).max()   # (what the kernel that produced `tiny` would have published)
    assert twin.tt_cpu_split_pairs_dual_parts(ptr(tiny), None, ptr(tr3), ptr(parts), ptr(S3), ptr(amx), M, N, 64, None, 0, None, None) == 0
    assert np.array_equal(tr3, tr) and S3[0] == S[0] and np.allclose(parts[0], tiny.sum(0), rtol=1e-5, atol=0)
    dw3, db3 = np.empty((N, K), np.float32), np.empty((N,), np.float32)
    assert twin.tt_cpu_linear_bwd_weight_pairs_tn_bias(ptr(tr3), ptr(xp), ptr(dw3), ptr(S3), N, K, M, None, 0, ptr(parts), 1, ptr(db3), None) == 0
    assert np.array_equal(dw3, dws) and np.allclose(db3, cs2, rtol=1e-6, atol=0)
    # attention on pairs against torch in fp64
    Fr, Nn, H = 2, 19, 2
    D = 64 * H
    qkv = rs.randn(Fr, Nn, 3 * D).astype(np.float32)
    qp = np.empty((Fr * Nn, 6 * D), np.uint16)
    assert twin.tt_cpu_split_pairs(ptr(qkv), ptr(qp), qkv.size, None, None) == 0
    of, lse, op = np.empty((Fr, Nn, D), np.float32), np.empty((Fr, H, Nn), np.float32), np.empty((Fr * Nn, 2 * D), np.uint16)
    assert twin.tt_cpu_attention_fwd_pairs(ptr(qp), ptr(op), ptr(of), ptr(lse), Fr, Nn, H, 64, 0.125, None) == 0
    q, k, v = torch.from_numpy(qkv).double().view(Fr, Nn, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) * 0.125
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(Fr, Nn, D).numpy()
    assert np.abs(of - ref).max() / np.abs(ref).max() < 1e-6 and np.abs(lse - torch.logsumexp(sc, -1).numpy()).max() < 1e-5
