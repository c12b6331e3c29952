"""The clip transforms of the reference (``video_transformations.py``) on GPU-resident frames.

Same class names, constructor arguments, defaults and - deliberately - the same random-number calls in the same order
(``random.random`` / ``random.uniform`` / ``random.randint`` / ``random.shuffle`` / ``torch.rand``), so that a run seeded like
the reference takes the same decisions and, because every kernel reproduces Pillow's arithmetic, produces the same tensors.
A clip is a ``torch.uint8`` tensor ``[fs, H, W, 3]`` on the device (what a video decoder hands over) instead of a list of
PIL images; ``ClipToTensor`` ends the chain with ``float32 [fs, 3, H, W]`` exactly as the reference does.

Reference quirks that are reproduced (and worth knowing):
* ``ColorJitter`` builds four adjustment closures, shuffles them, and then applies EACH to the ORIGINAL image, keeping only
  the last result (``video_transformations.py:774-777``): one randomly chosen adjustment takes effect, not a chain of four.
* ``RandomHorizontalFlip`` without annotations calls its helper with the default ``chance=0.5``, and ``0.5 < p`` is False
  for the default ``p=0.5`` (``:168-170,190-193``): training clips are never flipped and no random number is drawn.
* ``RandomGaussianBlur`` draws a fresh radius per frame inside the list comprehension (``:640``).
* ``RandomGrayscale`` draws from torch's generator, everything else from Python's ``random``.
Not built: the numpy-array (cv2 / skimage) code paths, ``RandomRotation``, ``RandomResize``, ``RandomVerticalFlip``, ``CenterCrop``
and the annotation-clip plumbing of the evaluation loaders.
"""
from __future__ import annotations

import math
import numbers
import random
from functools import lru_cache

import numpy as np
import torch

from . import hip_ops as ops

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=4096)
def resample_coeffs(in_size: int, out_size: int):
    """Pillow's bilinear taps for resizing a line of ``in_size`` pixels to ``out_size`` (Resample.c precompute_coeffs +
    normalize_coeffs_8bpc): (int32 [out_size, ksize], int32 [out_size, 2] = first input index and tap count) as CPU tensors.
    Vectorised over the output positions; every float64 operation and the left-to-right tap sum are the C loop's."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    cnt = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    a = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where((a < 1.0) & (x < cnt[:, None]), 1.0 - a, 0.0)
    ww = np.zeros(out_size, np.float64)
    for t in range(ksize):  # sequential sum, as the C loop
        ww = ww + w[:, t]
    v = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.trunc(0.5 + v * (1 << PRECISION_BITS)).astype(np.int32)
    kk = np.where(x < cnt[:, None], kk, 0).astype(np.int32)
    bounds = np.stack([xmin, cnt], axis=1).astype(np.int32)
    return torch.from_numpy(np.ascontiguousarray(kk)), torch.from_numpy(np.ascontiguousarray(bounds))


_DEVICE_COEFFS: dict = {}


def _coeffs_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    hit = _DEVICE_COEFFS.get(key)
    if hit is None:
        if len(_DEVICE_COEFFS) > 4096:
            _DEVICE_COEFFS.clear()
        kk, bounds = resample_coeffs(in_size, out_size)
        hit = _DEVICE_COEFFS[key] = (kk.to(device), bounds.to(device))
    return hit


def resized_crop(clip: torch.Tensor, i: int, j: int, h: int, w: int, size, to_tensor=None, flip: bool = False) -> torch.Tensor:
    """``img.crop((j, i, j + w, i + h)).resize((size[1], size[0]), BILINEAR)`` for every frame; with ``to_tensor=(mean, std)`` the
    result leaves as the normalised float tensor (optionally flipped) in the same launch."""
    OH, OW = size
    dev = clip.device
    x = clip
    y0 = i
    if OW != w:
        kk, bounds = _coeffs_on(dev, w, OW)
        x = ops.img_resample_h(x, kk, bounds, i, j, h)          # rows i..i+h only, columns j..j+w
        y0 = 0
    elif j != 0 or w != clip.shape[2]:
        x = x[:, :, j:j + w].contiguous()
    if OH != h or to_tensor is not None or y0 != 0 or x.shape[1] != h:
        kk, bounds = _coeffs_on(dev, h, OH)                      # (identity taps when OH == h)
        return ops.img_resample_v(x, kk, bounds, y0, to_tensor, flip)
    return x


def get_resize_sizes(im_h, im_w, size):
    if im_w < im_h:
        return int(size * im_h / im_w), size
    return size, int(size * im_w / im_h)


def resize_clip(clip: torch.Tensor, size, interpolation="bilinear") -> torch.Tensor:
    """``video_transformations.resize_clip`` (:56-94) for PIL clips: a number resizes the SHORTER side, keeping the aspect."""
    if interpolation != "bilinear":
        raise NotImplementedError("only bilinear resizing of image clips is built")
    _, im_h, im_w, _ = clip.shape
    if isinstance(size, numbers.Number):
        if (im_w <= im_h and im_w == size) or (im_h <= im_w and im_h == size):
            return clip
        new_h, new_w = get_resize_sizes(im_h, im_w, size)
    else:
        new_h, new_w = size[0], size[1]
    return resized_crop(clip, 0, 0, im_h, im_w, (new_h, new_w))


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, data_clip, annotation_clip=None):
        if annotation_clip is not None:
            raise NotImplementedError("annotation clips are not part of this build")
        ts = self.transforms
        k = 0
        while k < len(ts):
            # RandomResizedCrop [-> RandomHorizontalFlip] -> ClipToTensor: the crop's vertical resampling pass writes the
            # normalised float tensor directly (same arithmetic, same random draws, one launch and one uint8 round trip less)
            if isinstance(ts[k], RandomResizedCrop):
                nxt = k + 1
                flipper = ts[nxt] if nxt < len(ts) and isinstance(ts[nxt], RandomHorizontalFlip) else None
                nxt += flipper is not None
                if nxt < len(ts) and isinstance(ts[nxt], ClipToTensor):
                    data_clip = ts[k](data_clip, _finish=(flipper, ts[nxt]))
                    k = nxt + 1
                    continue
            data_clip = ts[k](data_clip)
            k += 1
        return data_clip


class RandomApply:
    def __init__(self, transforms, p=0.5):
        self.transforms, self.p = transforms, p

    def __call__(self, clip):
        if random.random() < self.p:
            for t in self.transforms:
                clip = t(clip)
        return clip


class ColorJitter:
    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0, per_frame=False):
        if per_frame:
            raise NotImplementedError("per_frame colour jitter is not used by the training pipeline")
        self.brightness, self.contrast, self.saturation, self.hue = brightness, contrast, saturation, hue

    def get_params(self, brightness, contrast, saturation, hue):
        b = random.uniform(max(0, 1 - brightness), 1 + brightness) if brightness > 0 else None
        c = random.uniform(max(0, 1 - contrast), 1 + contrast) if contrast > 0 else None
        s = random.uniform(max(0, 1 - saturation), 1 + saturation) if saturation > 0 else None
        h = random.uniform(-hue, hue) if hue > 0 else None
        return b, c, s, h

    def __call__(self, clip):
        b, c, s, h = self.get_params(self.brightness, self.contrast, self.saturation, self.hue)
        todo = []  # the reference's order of appends: brightness, saturation, hue, contrast (:762-769)
        if b is not None:
            todo.append((ops.IMG_BRIGHTNESS, b))
        if s is not None:
            todo.append((ops.IMG_SATURATION, s))
        if h is not None:
            todo.append((ops.IMG_HUE, h))
        if c is not None:
            todo.append((ops.IMG_CONTRAST, c))
        random.shuffle(todo)
        if not todo:
            raise UnboundLocalError("jittered_img")  # what the reference does with all four strengths at 0
        mode, value = todo[-1]  # each closure is applied to the ORIGINAL frame; only the last result is kept
        out = clip.clone()
        if mode == ops.IMG_HUE:
            return ops.img_color_(out, mode, 1.0, int(value * 255) % 256)
        return ops.img_color_(out, mode, value)


class RandomGrayscale:
    def __init__(self, p=0.2, per_frame=False):
        if per_frame:
            raise NotImplementedError("per_frame grayscale is not used by the training pipeline")
        self.p = p

    def __call__(self, clip):
        if torch.rand(1) < self.p:
            return ops.img_color_(clip.clone(), ops.IMG_GRAYSCALE)
        return clip


def gaussian_box_params(radius: float, passes: int = 3):
    """BoxBlur.c: the extended-box radius that approximates a Gaussian of std ``radius`` in ``passes`` passes, and its
    fixed-point weights (integer radius, ww, fw); float32 arithmetic as in the C source."""
    f32 = np.float32
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32((2 * l + 1) * (l * (l + 1) - 3 * sigma2))
    a = f32(a / f32(6 * (sigma2 - (l + 1) * (l + 1))))
    fr = f32(l + a)
    r = int(fr)
    ww = int(np.uint32(f32(1 << 24) / f32(fr * f32(2) + f32(1))))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def gaussian_blur(frames: torch.Tensor, radius: float) -> torch.Tensor:
    """``img.filter(ImageFilter.GaussianBlur(radius))`` on every frame of ``frames`` (all with the same radius)."""
    r, ww, fw = gaussian_box_params(radius)
    out = frames
    for direction in (0, 1):
        for _ in range(3):
            out = ops.img_box_blur(out, direction, r, ww, fw)
    return out


class RandomGaussianBlur:
    def __init__(self, p=0.5, radius_min=0.1, radius_max=2., per_frame=False):
        if per_frame:
            raise NotImplementedError("per_frame blur is not used by the training pipeline")
        self.p, self.radius_min, self.radius_max = p, radius_min, radius_max

    def __call__(self, clip):
        if random.random() < self.p:
            radii = [random.uniform(self.radius_min, self.radius_max) for _ in range(clip.shape[0])]  # one draw per frame (:640)
            return torch.cat([gaussian_blur(clip[t:t + 1].contiguous(), radii[t]) for t in range(clip.shape[0])], dim=0)
        return clip


class Resize:
    def __init__(self, size, interpolation="bilinear"):
        self.size, self.interpolation = size, interpolation

    def __call__(self, data_clip, annotaion_clip=None):
        if annotaion_clip is not None:
            raise NotImplementedError("annotation clips are not part of this build")
        return resize_clip(data_clip, self.size, self.interpolation)


class RandomResizedCrop:
    def __init__(self, size, scale=(0.4, 1.0), ratio=(3. / 4., 4. / 3.), interpolation="bilinear"):
        self.size = size if isinstance(size, (tuple, list)) else (size, size)
        self.interpolation, self.scale, self.ratio = interpolation, scale, ratio

    @staticmethod
    def get_params(clip, scale, ratio):
        """(i, j, h, w) with the reference's draws (``video_transformations.py:447-489``)."""
        height, width = clip.shape[1], clip.shape[2]
        area = height * width
        for _ in range(10):
            target_area = random.uniform(*scale) * area
            log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
            aspect_ratio = math.exp(random.uniform(*log_ratio))
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = random.randint(0, height - h)
                j = random.randint(0, width - w)
                return i, j, h, w
        in_ratio = float(width) / float(height)
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def __call__(self, data_clip, annotaion_clip=None, _finish=None):
        if annotaion_clip is not None:
            raise NotImplementedError("annotation clips are not part of this build")
        i, j, h, w = self.get_params(data_clip, self.scale, self.ratio)
        if _finish is None:
            return resized_crop(data_clip, i, j, h, w, self.size)
        flipper, to_tensor = _finish
        flip = flipper is not None and flipper.will_flip()
        return resized_crop(data_clip, i, j, h, w, self.size, to_tensor=to_tensor.mean_std(), flip=flip)


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, data_clip, annotation_clip=None):
        if annotation_clip is not None:
            raise NotImplementedError("annotation clips are not part of this build")
        return torch.flip(data_clip, dims=[2]) if self.will_flip() else data_clip

    def will_flip(self) -> bool:
        chance = 0.5  # the helper's default argument: no random number is drawn on this path
        return chance < self.p


class ClipToTensor:
    """uint8 [fs, H, W, 3] -> float32 [fs, 3, H, W] in [0, 1], normalised when mean and std are given."""

    def __init__(self, mean=None, std=None):
        self.mean, self.std = mean, std

    def __call__(self, data_clip, annotation_clip=None):
        if annotation_clip is not None:
            raise NotImplementedError("annotation clips are not part of this build")
        _, H, W, _ = data_clip.shape
        return resized_crop(data_clip, 0, 0, H, W, (H, W), to_tensor=self.mean_std())

    def mean_std(self):
        if self.mean is not None and self.std is not None:
            return self.mean, self.std
        return (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)


def training_transforms(input_resolution: int = 224):
    """The two ``Compose`` objects ``time_tuning.time_tuning`` builds (``time_tuning.py:588-593``): (frame_transform, video_transform)."""
    rand_color_jitter = RandomApply([ColorJitter(brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2)], p=0.8)
    data_transform = Compose([rand_color_jitter, RandomGrayscale(), RandomGaussianBlur()])
    video_transform = Compose([Resize(input_resolution), RandomResizedCrop((input_resolution, input_resolution)), RandomHorizontalFlip(),
                               ClipToTensor(mean=[0.485, 0.456, 0.406], std=[0.228, 0.224, 0.225])])
    return data_transform, video_transform
