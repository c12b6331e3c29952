"""The coarse entry points of the C ABI (tt_vit_forward, tt_mlp_head_forward, tt_scores_sinkhorn, tt_adamw_ema_step; SURVEY.md
8(b)): their plain-C twins against torch fp64 on the CPU; on the GPU the HIP library against the twins through one call site,
and - the claim they are built on - BIT-identical results with the fine-grained launch sequence they replace."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_twin, timet_oracle as O
from timetuning_amd import _lib as L


def _re(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _bf(a):
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16).double().numpy()


class _Mem:
    """Places numpy arrays where a library expects them: host pointers for the twin, device copies for HIP; ``back`` copies an
    output array's device copy back."""

    def __init__(self, device=None):
        self.device, self.held = device, {}

    def __call__(self, a):
        if a is None:
            return None
        if self.device is None:
            self.held[id(a)] = (a, a)
            return a.ctypes.data_as(C.c_void_p).value
        t = torch.from_numpy(a).to(self.device)
        self.held[id(a)] = (a, t)
        return t.data_ptr()

    def back(self, *arrays):
        if self.device is not None:
            torch.cuda.synchronize()
            for a in arrays:
                a[...] = self.held[id(a)][1].cpu().numpy()

    @property
    def stream(self):
        return torch.cuda.current_stream().cuda_stream if self.device else None


def _vit_case(seed=3, D=128, heads=2, hidden=256, depth=2, patch=16, C_=3, H=32, W=48, n_src=4):
    rng = np.random.default_rng(seed)
    f32 = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)
    N = 1 + (H // patch) * (W // patch)
    w = dict(patch_w=f32(D, C_ * patch * patch, scale=0.05), patch_b=f32(D, scale=0.1), cls=f32(D), pos=f32(N, D, scale=0.3),
             norm_w=1 + 0.1 * f32(D), norm_b=0.1 * f32(D), blocks=[])
    for _ in range(depth):
        w["blocks"].append(dict(norm1_w=1 + 0.1 * f32(D), norm1_b=0.1 * f32(D), qkv_w=f32(3 * D, D, scale=0.08), qkv_b=f32(3 * D, scale=0.1),
                                proj_w=f32(D, D, scale=0.08), proj_b=f32(D, scale=0.1), norm2_w=1 + 0.1 * f32(D), norm2_b=0.1 * f32(D),
                                fc1_w=f32(hidden, D, scale=0.08), fc1_b=f32(hidden, scale=0.1), fc2_w=f32(D, hidden, scale=0.08),
                                fc2_b=f32(D, scale=0.1)))
    img = f32(n_src, C_, H, W)
    fmap = np.array([2, 0, 3], np.int32)
    return dict(w=w, img=img, fmap=fmap, D=D, heads=heads, hidden=hidden, patch=patch, C=C_, H=H, W=W, N=N, F=len(fmap))


def _run_vit(lib, prefix, mem, case, planes, drop_cls=False, want_qkv=False, want_probs=False, from_tokens=None):
    """prefix + vit_forward on ``case``; returns (tokens, normed, qkv, probs) as numpy."""
    w, D, N, Fr = case["w"], case["D"], case["N"], case["F"]
    nb = len(w["blocks"])
    arr = (L.VitBlockParams * nb)()
    split = getattr(lib, prefix + "split_planes")
    split_pairs = getattr(lib, prefix + "split_pairs")
    for j, b in enumerate(w["blocks"]):
        for k, v in b.items():
            setattr(arr[j], k, mem(v))
        if planes:
            for k in ("qkv_w", "proj_w", "fc1_w", "fc2_w"):
                wp = np.empty((planes,) + b[k].shape, np.uint16)   # (planes == 2: fp16 pairs [out][2 in] - the same number of bytes)
                src, dst = mem(b[k]), mem(wp)
                if planes == 2:
                    assert split_pairs(src, dst, b[k].size, None, mem.stream) == 0
                else:
                    assert split(src, dst, b[k].size, planes, b[k].size, mem.stream) == 0
                setattr(arr[j], k + "p", dst)
    vp = L.VitParams()
    vp.patch_w, vp.patch_b, vp.cls, vp.pos = mem(w["patch_w"]), mem(w["patch_b"]), mem(w["cls"]), mem(w["pos"])
    vp.blocks, vp.n_blocks, vp.norm_w, vp.norm_b = arr, nb, mem(w["norm_w"]), mem(w["norm_b"])
    vp.dim, vp.heads, vp.hidden, vp.patch, vp.planes = D, case["heads"], case["hidden"], case["patch"], planes
    if planes == 2 and w["patch_w"].size // D % 32 == 0:   # prepare_tokens on pairs too (where tt_vit_forward's shape rules hold; else it falls back)
        pwp = np.empty((D, 2 * (w["patch_w"].size // D)), np.uint16)
        dst = mem(pwp)
        assert split_pairs(vp.patch_w, dst, w["patch_w"].size, None, mem.stream) == 0   # (the copy mem() made above: a second mem() of it would free that one)
        vp.patch_wp = dst
    tokens = np.empty((Fr, N, D), np.float32) if from_tokens is None else from_tokens.copy()
    normed = np.empty((Fr * (N - 1), D) if drop_cls else (Fr, N, D), np.float32)
    qkv = np.empty((Fr, N, 3 * D), np.float32) if want_qkv else None
    probs = np.empty((Fr, case["heads"], N, N), np.float32) if want_probs else None
    nbytes = getattr(lib, prefix + "vit_forward_workspace_bytes")(Fr, N, D, case["hidden"], planes)
    ws = np.empty(max(nbytes, 16), np.uint8)
    img = None if from_tokens is not None else case["img"]
    rc = getattr(lib, prefix + "vit_forward")(C.byref(vp), mem(img), mem(case["fmap"]) if img is not None else None, Fr, case["C"], case["H"],
                                              case["W"], mem(tokens), mem(normed), int(drop_cls), mem(qkv), mem(probs), mem(ws), nbytes, mem.stream)
    assert rc == 0, (rc, lib.tt_last_error() if hasattr(lib, "tt_last_error") else "")
    mem.back(*[a for a in (tokens, normed, qkv, probs) if a is not None])
    return tokens, normed, qkv, probs


def _vit_torch(case, from_tokens=None):
    """fp64 torch restatement of dino_vision_transformer.py:236-252 + :135-153 + final norm."""
    w, D, heads, patch = case["w"], case["D"], case["heads"], case["patch"]
    t = lambda a: torch.from_numpy(a).double()
    if from_tokens is None:
        x = F.conv2d(t(case["img"][case["fmap"]]), t(w["patch_w"]).view(D, case["C"], patch, patch), t(w["patch_b"]), stride=patch)
        x = torch.cat([t(w["cls"]).expand(x.shape[0], 1, D), x.flatten(2).transpose(1, 2)], 1) + t(w["pos"])
    else:
        x = t(from_tokens)
    qkv = probs = None
    for b in w["blocks"]:
        h = F.layer_norm(x, (D,), t(b["norm1_w"]), t(b["norm1_b"]), 1e-6)
        qkv = F.linear(h, t(b["qkv_w"]), t(b["qkv_b"]))
        Fr, N, _ = qkv.shape
        q, k, v = qkv.view(Fr, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        probs = torch.softmax(q @ k.transpose(-1, -2) * (D // heads) ** -0.5, -1)
        x = x + F.linear((probs @ v).transpose(1, 2).reshape(Fr, N, D), t(b["proj_w"]), t(b["proj_b"]))
        h = F.layer_norm(x, (D,), t(b["norm2_w"]), t(b["norm2_b"]), 1e-6)
        x = x + F.linear(F.gelu(F.linear(h, t(b["fc1_w"]), t(b["fc1_b"]))), t(b["fc2_w"]), t(b["fc2_b"]))
    normed = F.layer_norm(x, (D,), t(w["norm_w"]), t(w["norm_b"]), 1e-6)
    return x.numpy(), normed.numpy(), qkv.numpy(), probs.numpy()


def _small_cases(lib, prefix, mem):
    """mlp_head_forward, scores_sinkhorn, adamw_ema_step on one side; returns inputs and results."""
    rng = np.random.default_rng(5)
    f32 = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)
    R = {}
    # projection head: Linear GELU Linear GELU Linear
    M, dims = 50, [48, 96, 64, 32]
    x = f32(M, dims[0])
    lay = [(f32(dims[i + 1], dims[i], scale=0.15), f32(dims[i + 1], scale=0.1)) for i in range(3)]
    arr = (L.LinearParams * 3)()
    for i, (w, b) in enumerate(lay):
        arr[i] = L.LinearParams(mem(w), mem(b), w.shape[0], w.shape[1])
    out = np.empty((M, dims[-1]), np.float32)
    nb = getattr(lib, prefix + "mlp_head_forward_workspace_bytes")(M, arr, 3)
    ws = np.empty(max(nb, 16), np.uint8)
    assert getattr(lib, prefix + "mlp_head_forward")(mem(x), M, arr, 3, mem(out), 0, mem(ws), nb, mem.stream) == 0
    mem.back(out)
    R["head"] = (x, lay, out)
    # scores + assignment, with queue rows
    B, Qr, K, dim = 60, 24, 10, 32
    z, queue, protos = f32(B, dim), f32(Qr, dim), f32(K, dim)
    protos /= np.linalg.norm(protos, axis=1, keepdims=True)
    for qrows in (Qr, 0):
        scores, q = np.empty((B + qrows, K), np.float32), np.empty((B, K), np.float32)
        nb = getattr(lib, prefix + "scores_sinkhorn_workspace_bytes")(B, qrows, K, dim)
        ws = np.empty(max(nb, 16), np.uint8)
        assert getattr(lib, prefix + "scores_sinkhorn")(mem(z), B, mem(queue) if qrows else None, qrows, mem(protos), K, dim, mem(scores), mem(q), B,
                                                        0.05, 10, 0, mem(ws), nb, mem.stream) == 0
        mem.back(scores, q)
        R[f"scores{qrows}"] = (z, queue[:qrows], protos, scores, q)
    # AdamW over 45 tensors (more than one table of TT_MAX_TENSORS) + prototype renormalisation + EMA
    K2, d2 = 6, 16
    sizes = [K2 * d2] + [int(s) for s in rng.integers(3, 70, 44)]
    ps = [f32(s) for s in sizes]
    gs = [f32(s, scale=0.1) for s in sizes]
    ms = [0.01 * f32(s) for s in sizes]
    vs = [np.abs(0.01 * f32(s)) for s in sizes]
    p0 = [p.copy() for p in ps]
    m0 = [m.copy() for m in ms]
    v0 = [v.copy() for v in vs]
    tab = (L.AdamwTensor * len(sizes))()
    for i in range(len(sizes)):
        tab[i] = L.AdamwTensor(mem(ps[i]), mem(gs[i]), mem(ms[i]), mem(vs[i]), sizes[i], 1e-2 if i % 2 else 3e-3, 0.04 if i % 3 else 0.0)
    n_flat = 301
    student, teacher, tprot = f32(n_flat), f32(n_flat), f32(K2, d2)
    teacher0, tprot0 = teacher.copy(), tprot.copy()
    assert getattr(lib, prefix + "adamw_ema_step")(tab, len(sizes), 3, 0.9, 0.999, 1e-8, tab[0].p, K2, d2, mem(teacher), mem(student), n_flat,
                                                   mem(tprot), 0.995, mem.stream) == 0
    mem.back(*ps, *ms, *vs, teacher, tprot)
    R["update"] = (sizes, p0, gs, m0, v0, ps, ms, vs, student, teacher0, teacher, tprot0, tprot, (K2, d2))
    return R


@pytest.fixture(scope="module")
def twin():
    return cpu_twin.load()


def test_twin_vit_forward_vs_torch(twin):
    case = _vit_case()
    tok_ref, normed_ref, qkv_ref, probs_ref = _vit_torch(case)
    tok, normed, qkv, probs = _run_vit(twin, "tt_cpu_", _Mem(), case, 0, want_qkv=True, want_probs=True)
    assert _re(tok, tok_ref) < 2e-6 and _re(normed, normed_ref) < 2e-6 and _re(qkv, qkv_ref) < 2e-6 and _re(probs, probs_ref) < 2e-6
    _, dropped, _, _ = _run_vit(twin, "tt_cpu_", _Mem(), case, 0, drop_cls=True)
    assert _re(dropped, normed_ref[:, 1:].reshape(-1, case["D"])) < 2e-6                  # get_features drops the cls token
    tok3, normed3, _, _ = _run_vit(twin, "tt_cpu_", _Mem(), case, 3)                        # three bf16 planes: fp32-accurate
    assert _re(tok3, tok_ref) < 5e-6 and _re(normed3, normed_ref) < 5e-6
    tok2, normed2, qkv2, _ = _run_vit(twin, "tt_cpu_", _Mem(), case, 2, want_qkv=True)      # fp16 pairs: fp32-accurate (fp32 attention route)
    assert _re(tok2, tok_ref) < 5e-6 and _re(normed2, normed_ref) < 5e-6 and _re(qkv2, qkv_ref) < 5e-6
    tok2, normed2, _, _ = _run_vit(twin, "tt_cpu_", _Mem(), case, 2)                        # ... and the pair attention route
    assert _re(tok2, tok_ref) < 5e-6 and _re(normed2, normed_ref) < 5e-6
    tok1, normed1, _, _ = _run_vit(twin, "tt_cpu_", _Mem(), case, 1)                        # the bf16 path (bf16 attention: hd = 64)
    assert _re(normed1, normed_ref) < 3e-2
    # continuing from a residual stream (img == NULL): the EMA teacher's form
    start = np.random.default_rng(1).standard_normal((case["F"], case["N"], case["D"])).astype(np.float32)
    tok_c, normed_c, _, _ = _run_vit(twin, "tt_cpu_", _Mem(), case, 0, from_tokens=start)
    ref_c = _vit_torch(case, from_tokens=start)
    assert _re(tok_c, ref_c[0]) < 2e-6 and _re(normed_c, ref_c[1]) < 2e-6


def _check_small(R):
    x, lay, out = R["head"]
    h = torch.from_numpy(x).double()
    for i, (w, b) in enumerate(lay):
        h = F.linear(h, torch.from_numpy(w).double(), torch.from_numpy(b).double())
        if i < 2:
            h = F.gelu(h)
    assert _re(out, h.numpy()) < 2e-5
    for key in ("scores24", "scores0"):
        z, queue, protos, scores, q = R[key]
        rows = np.concatenate([z, queue]) if len(queue) else z
        ref_scores = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float64) @ protos.astype(np.float64).T
        assert _re(scores, ref_scores) < 2e-5
        ref_q = O.sinkhorn(torch.exp(torch.from_numpy(ref_scores) / 0.05).t(), 10).numpy()[: len(z)]        # my_utils.py:246-274, fp64
        assert _re(q, ref_q) < 2e-4, key
    sizes, p0, gs, m0, v0, ps, ms, vs, student, teacher0, teacher, tprot0, tprot, (K2, d2) = R["update"]
    for i in range(len(sizes)):
        p = torch.nn.Parameter(torch.from_numpy(p0[i]).double())
        p.grad = torch.from_numpy(gs[i]).double()
        opt = torch.optim.AdamW([p], lr=1e-2 if i % 2 else 3e-3, weight_decay=0.04 if i % 3 else 0.0)
        opt.state[p] = dict(step=torch.tensor(2.0), exp_avg=torch.from_numpy(m0[i]).double(), exp_avg_sq=torch.from_numpy(v0[i]).double())
        opt.step()
        ref = p.detach().numpy()
        if i == 0:   # the prototypes: renormalised rows
            ref = ref.reshape(K2, d2)
            ref = (ref / np.linalg.norm(ref, axis=1, keepdims=True)).reshape(-1)
            protos_new = ref
        assert _re(ps[i], ref) < 2e-6, i
        assert _re(ms[i], opt.state[p]["exp_avg"].numpy()) < 2e-6 and _re(vs[i], opt.state[p]["exp_avg_sq"].numpy()) < 2e-6
    assert _re(teacher, teacher0.astype(np.float64) * (1 - 0.995) + student.astype(np.float64) * 0.995) < 2e-6
    tp = tprot0.astype(np.float64) * (1 - 0.995) + protos_new.reshape(K2, d2) * 0.995
    assert _re(tprot, tp / np.linalg.norm(tp, axis=1, keepdims=True)) < 2e-6


def test_twin_head_scores_update_vs_torch(twin):
    _check_small(_small_cases(twin, "tt_cpu_", _Mem()))


# ---- GPU ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_hip_coarse_entries_equal_their_twins(twin):
    lib = L.load()
    case = _vit_case()
    a = _run_vit(lib, "tt_", _Mem("cuda"), case, 2)   # fp16 pairs without the last block's qkv: the pair attention kernel's route
    b = _run_vit(twin, "tt_cpu_", _Mem(), case, 2)
    assert _re(a[0], b[0]) < 2e-5 and _re(a[1], b[1]) < 2e-5
    case8 = _vit_case(patch=8)                         # C P P = 192 <= 3 D: prepare_tokens on pair operands as well (tt_patch_embed_fwd_pairs)
    a = _run_vit(lib, "tt_", _Mem("cuda"), case8, 2)
    b = _run_vit(twin, "tt_cpu_", _Mem(), case8, 2)
    assert _re(a[0], b[0]) < 2e-5 and _re(a[1], b[1]) < 2e-5
    for planes, tol in ((0, 2e-5), (3, 2e-5), (2, 2e-5), (1, 3e-2)):
        a = _run_vit(lib, "tt_", _Mem("cuda"), case, planes, want_qkv=planes != 1, want_probs=planes == 0)
        b = _run_vit(twin, "tt_cpu_", _Mem(), case, planes, want_qkv=planes != 1, want_probs=planes == 0)
        for x, y in zip(a, b):
            if x is not None:
                assert _re(x, y) < tol, planes
    a = _run_vit(lib, "tt_", _Mem("cuda"), case, 0, drop_cls=True)
    b = _run_vit(twin, "tt_cpu_", _Mem(), case, 0, drop_cls=True)
    assert _re(a[1], b[1]) < 2e-5
    R = _small_cases(lib, "tt_", _Mem("cuda"))
    _check_small(R)
    T = _small_cases(twin, "tt_cpu_", _Mem())
    assert _re(R["head"][2], T["head"][2]) < 2e-5 and _re(R["scores24"][4], T["scores24"][4]) < 2e-4
    for i in range(len(R["update"][0])):
        assert _re(R["update"][5][i], T["update"][5][i]) < 2e-6


@pytest.mark.gpu
def test_coarse_entry_argument_checks():
    lib = L.load()
    vp = L.VitParams()
    vp.dim, vp.heads, vp.hidden, vp.patch, vp.planes, vp.n_blocks = 128, 2, 256, 16, 4, 0
    t = torch.zeros(1, 7, 128, device="cuda")
    assert lib.tt_vit_forward(C.byref(vp), None, None, 1, 3, 32, 48, t.data_ptr(), None, 0, None, None, None, 0, None) == -1
    assert b"planes" in lib.tt_last_error()
    vp.planes = 0
    assert lib.tt_vit_forward(C.byref(vp), None, None, 1, 3, 30, 48, t.data_ptr(), None, 0, None, None, None, 0, None) == -1
    assert lib.tt_vit_forward(C.byref(vp), None, None, 1, 3, 32, 48, t.data_ptr(), None, 0, None, None, None, 0, None) == 0   # nothing to do
    arr = (L.LinearParams * 2)(L.LinearParams(t.data_ptr(), None, 8, 16), L.LinearParams(t.data_ptr(), None, 4, 9))
    assert lib.tt_mlp_head_forward(t.data_ptr(), 4, arr, 2, t.data_ptr(), 0, t.data_ptr(), 1 << 20, None) == -1
    assert b"layer 1 takes 9" in lib.tt_last_error()
    assert lib.tt_scores_sinkhorn(t.data_ptr(), 4, None, 0, t.data_ptr(), 4, 8, t.data_ptr(), t.data_ptr(), 9, 0.05, 3, 0, t.data_ptr(), 1 << 20, None) == -1
    # the bf16 patch embedding: shape rules and the workspace size are checked, nothing is launched on a bad call
    big = torch.zeros(1 << 20, device="cuda")
    p_ = big.data_ptr()
    assert lib.tt_patch_embed_planes_workspace_bytes(2, 3, 32, 48, 16) == 2 * 7 * 768 * 2 + lib.tt_linear_ksplit_workspace_bytes()   # rows + the K-split block (ABI 7)
    assert lib.tt_patch_embed_fwd_planes(p_, None, p_, p_, p_, p_, p_, 2, 3, 32, 48, 16, 100, p_, 1 << 22, None) == -1 and b"D % 64" in lib.tt_last_error()
    assert lib.tt_patch_embed_fwd_planes(p_, None, p_, p_, p_, p_, p_, 2, 3, 32, 48, 16, 128, p_, 100, None) == -1 and b"workspace too small" in lib.tt_last_error()
    assert lib.tt_patch_embed_fwd_planes(p_, None, p_, p_, p_, p_, p_, 2, 3, 30, 48, 16, 128, p_, 1 << 22, None) == -1
    assert lib.tt_patch_embed_fwd_planes(p_, None, None, p_, p_, p_, p_, 2, 3, 32, 48, 16, 128, p_, 1 << 22, None) == -1


def _tiny_model(teacher=False, queue=0, K=12):
    from timetuning_amd import synth
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer, TimeT

    torch.manual_seed(0)
    fe = FeatureExtractor("dino-s16", "", [256, 128, 64], unfreeze_layers=["blocks.11", "blocks.10"],
                          vit_cfg=dict(embed_dim=128, depth=12, num_heads=2, patch_size=16), init="stress")
    model = TimeT(fe, K, prototype_init=torch.from_numpy(synth.make_prototypes(K, fe.feature_dim))).cuda()
    opt = SwavOptimizer(model, "AdamW", True, 1e-4, 1e-3, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 8), 8, 1)
    if teacher:
        model.init_momentum_teacher()
        model.set_momentum_teacher_schedular_params(0.99, 1.0, 1, 8)
    if queue:
        model.init_queue(queue)
    return model, opt


class _fine_grained:
    """Forces the launch sequences through the op-level entry points (what bench.py's per-GEMM profile does)."""

    def __enter__(self):
        from timetuning_amd import hip_ops
        hip_ops.PROFILE = []

    def __exit__(self, *a):
        from timetuning_amd import hip_ops
        hip_ops.PROFILE = None


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16x3", "bf16x6", "bf16", "bf16x3"])
def test_coarse_path_is_bit_identical_to_the_fine_grained_sequence(mode):
    """The same kernels in the same order: extractor outputs, the loss, every gradient, and the state after two full training
    iterations (teacher + queue) must agree BIT FOR BIT between the coarse entry points and the one-call-per-op sequence."""
    from timetuning_amd import hip_ops as ops, synth

    ops.set_gemm_precision(mode)
    keep, ops.PAIRS_MIN_ROWS = ops.PAIRS_MIN_ROWS, 0   # (the pair kernels at this tiny size too)
    try:
        x = torch.from_numpy(synth.make_clips(2, 3, 224, seed=4)).cuda()
        results = []
        for fine in (False, True):
            model, opt = _tiny_model(teacher=True, queue=392 * 2)
            out = {}
            ctx = _fine_grained() if fine else None
            if ctx: ctx.__enter__()
            try:
                with torch.no_grad():
                    f, attn = model.feature_extractor(x.view(6, 3, 224, 224))
                    q, sc = model.similarity(f[:2])
                out.update(f=f, attn=attn, q=q, sc=sc)
                for it in range(2):
                    perm = torch.from_numpy(np.random.default_rng(it).permutation(2 * 196))
                    loss = model.get_loss(x, queue_perm=perm)
                    if fine:
                        opt.step(loss)
                        model.normalize_prototypes()
                        model.update_momentum_teacher(opt.global_step)
                    else:
                        model.train_update(opt, loss, opt.global_step + 1)
                    out[f"loss{it}"] = loss.detach().clone()
                with torch.no_grad():   # full queue by now: the queue rows take part in the assignment
                    out["q_full"], _ = model.similarity(f[:2])
            finally:
                if ctx: ctx.__exit__()
            out.update({"p." + k: v.detach().clone() for k, v in model.state_dict().items()})
            results.append(out)
        a, b = results
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (mode, k, (a[k].float() - b[k].float()).abs().max().item())
    finally:
        ops.PAIRS_MIN_ROWS = keep
        ops.set_gemm_precision("f32")
