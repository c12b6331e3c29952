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
