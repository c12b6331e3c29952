"""CPU oracle for the clip input pipeline.  TEST INFRASTRUCTURE ONLY.

The reference's training transforms (``video_transformations.py``, wired at ``time_tuning.py:588-593``) operate on lists
of PIL images and delegate every pixel operation to Pillow (directly, or through torchvision's PIL backend).  Pillow is a
third-party dependency whose sources are not under ``/root/reference``; it IS installed in the build image (12.2.0), so the
NumPy restatements below are pinned the strong way: ``tests/test_image_ops_oracle.py`` checks each of them bit for bit
against Pillow itself, and ``oracle/gen_golden.py`` records outputs of the reference's own ``Compose`` pipelines.

Restated algorithms (Pillow's C sources, by file):
  resize_bilinear      Resample.c: separable convolution, triangle filter with support scaled by the down-sampling factor,
                       coefficients in 22-bit fixed point, uint8 rounding after EACH of the two passes (horizontal first)
  to_gray              Convert.c rgb2l: (19595 R + 38470 G + 7471 B + 0x8000) >> 16
  blend                Blend.c: in1 + alpha (in2 - in1) in float32, truncated (alpha in [0,1]) or clipped
  enhance_*            ImageEnhance: blend against black / the mean-gray image / the grayscale image
  rgb2hsv, hsv2rgb     Convert.c (float / double mix reproduced), hue shift as torchvision's adjust_hue (uint8 wrap-around)
  gaussian_blur        BoxBlur.c: three extended box-blur passes per direction, 24-bit fixed point, uint8 rounding per pass
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
PRECISION_BITS = 32 - 8 - 2


# ---- Resample.c ---------------------------------------------------------------------------------

def resample_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the whole input [0, in_size).
    Returns (kk int32 [out_size, ksize], bounds int32 [out_size, 2] = (first input index, tap count))."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    kk, bounds = resample_coeffs(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc = acc + src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize_bilinear(img: np.ndarray, size) -> np.ndarray:
    """``Image.resize((w, h), BILINEAR)`` on a uint8 [H, W, C] array."""
    ow, oh = size
    out = img
    if ow != img.shape[1]:
        out = _resample_axis(out, ow, 1)
    if oh != img.shape[0]:
        out = _resample_axis(out, oh, 0)
    return out


def get_resize_sizes(im_h, im_w, size):
    """video_transformations.py:97-104."""
    if im_w < im_h:
        return int(size * im_h / im_w), size
    return size, int(size * im_w / im_h)


# ---- Convert.c / Blend.c / ImageEnhance -----------------------------------------------------------

def to_gray(a: np.ndarray) -> np.ndarray:
    a = a.astype(np.int64)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def gray3(a: np.ndarray) -> np.ndarray:
    return np.repeat(to_gray(a)[..., None], 3, axis=-1)


def blend(in1: np.ndarray, in2: np.ndarray, alpha: float) -> np.ndarray:
    alpha = f32(alpha)
    t = in1.astype(f32) + alpha * (in2.astype(np.int32) - in1.astype(np.int32)).astype(f32)
    if 0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def enhance_brightness(a, factor):
    return blend(np.zeros_like(a), a, factor)


def enhance_contrast(a, factor):
    g = to_gray(a)
    mean = int(float(g.astype(np.int64).sum()) / g.size + 0.5)
    return blend(np.full_like(a, mean), a, factor)


def enhance_saturation(a, factor):
    return blend(gray3(a), a, factor)


def rgb2hsv(a: np.ndarray) -> np.ndarray:
    r, g, b = [a[..., i].astype(np.int32) for i in range(3)]
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    safe = np.where(cr == 0, f32(1), cr)
    s = (cr / np.where(maxc == 0, 1, maxc).astype(f32)).astype(f32)
    rc, gc, bc = [((maxc - c).astype(f32) / safe).astype(f32) for c in (r, g, b)]
    h = np.where(r == maxc, (bc - gc).astype(np.float64),
                 np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc, 4.0 + gc.astype(np.float64) - rc)).astype(f32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv2rgb(a: np.ndarray) -> np.ndarray:
    h, s, v = [a[..., i].astype(np.int32) for i in range(3)]
    hf = h.astype(f32) * f32(6.0) / f32(255.0)
    i = np.floor(hf).astype(np.int32)
    f = hf - i.astype(f32)
    fs = s.astype(f32) / f32(255.0)
    fv = v.astype(f32)

    def cround(x):  # C round(): halves away from zero
        x = x.astype(np.float64)
        return np.clip(np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int32), 0, 255)

    p, q, t = cround(fv * (f32(1.0) - fs)), cround(fv * (f32(1.0) - fs * f)), cround(fv * (f32(1.0) - fs * (f32(1.0) - f)))
    i6 = i % 6
    r = np.choose(i6, [v, q, p, p, t, v])
    g = np.choose(i6, [t, v, v, q, p, p])
    b = np.choose(i6, [p, p, t, v, v, q])
    out = np.where((s == 0)[..., None], np.stack([v, v, v], -1), np.stack([r, g, b], -1))
    return out.astype(np.uint8)


def hue_shift_u8(hue_factor: float) -> int:
    """torchvision adjust_hue (PIL backend): ``np_h += np.uint8(hue_factor * 255)`` with uint8 wrap-around."""
    return int(hue_factor * 255) % 256


def adjust_hue(a: np.ndarray, hue_factor: float) -> np.ndarray:
    hsv = rgb2hsv(a)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_u8(hue_factor)) % 256
    return hsv2rgb(hsv)


# ---- BoxBlur.c ------------------------------------------------------------------------------------

def gaussian_box_radius(radius: float, passes: int = 3) -> np.float32:
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32((2 * l + 1) * (l * (l + 1) - 3 * sigma2))
    a = f32(a / f32(6 * (sigma2 - (l + 1) * (l + 1))))
    return f32(l + a)


def box_weights(fr: np.float32):
    """(integer radius, ww, fw) of ImagingHorizontalBoxBlur."""
    radius = int(fr)
    ww = int(np.uint32(f32(1 << 24) / f32(fr * f32(2) + f32(1))))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def _hbox(img: np.ndarray, fr: np.float32) -> np.ndarray:
    radius, ww, fw = box_weights(fr)
    W = img.shape[1]
    x = np.arange(W)
    im = img.astype(np.int64)
    acc = np.zeros_like(im)
    for d in range(-radius, radius + 1):
        acc += im[:, np.clip(x + d, 0, W - 1)]
    far = im[:, np.clip(x - radius - 1, 0, W - 1)] + im[:, np.clip(x + radius + 1, 0, W - 1)]
    return ((acc * ww + far * fw + (1 << 23)) >> 24).astype(np.uint8)


def gaussian_blur(img: np.ndarray, radius: float) -> np.ndarray:
    """``img.filter(ImageFilter.GaussianBlur(radius))`` on a uint8 [H, W, C] array."""
    fr = gaussian_box_radius(radius)
    out = img
    for _ in range(3):
        out = _hbox(out, fr)
    out = out.transpose(1, 0, 2)
    for _ in range(3):
        out = _hbox(out, fr)
    return out.transpose(1, 0, 2)


# ---- the tail of the pipeline: crop -> resize -> (flip) -> ClipToTensor(mean, std) -------------------

def resized_crop_to_tensor(img: np.ndarray, crop, size, flip: bool, mean, std) -> np.ndarray:
    """crop = (i, j, h, w) as RandomResizedCrop.get_params returns it; size = (H_out, W_out).  Returns float32 [3, H_out, W_out]
    = (ToTensor(resized) - mean) / std  (video_transformations.py:491-500,168-179,262-276)."""
    i, j, h, w = crop
    out = resize_bilinear(img[i:i + h, j:j + w], (size[1], size[0]))
    if flip:
        out = out[:, ::-1]
    t = np.ascontiguousarray(out.transpose(2, 0, 1)).astype(f32) / f32(255)
    return ((t - np.asarray(mean, f32)[:, None, None]) / np.asarray(std, f32)[:, None, None]).astype(f32)
