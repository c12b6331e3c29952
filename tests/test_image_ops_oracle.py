"""The input-pipeline oracle (oracle/image_ops.py) against Pillow itself, bit for bit.  Pillow is the third-party library the
reference's transforms delegate to (video_transformations.py); it is installed in the build image, so the restatements are
pinned against the real thing rather than against fixtures."""
import numpy as np
import pytest

from oracle import image_ops as I

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter, ImageStat  # noqa: E402


def rnd(seed, *shape):
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


@pytest.mark.parametrize("h,w,ow,oh", [(100, 160, 224, 224), (300, 500, 224, 224), (333, 257, 224, 298), (50, 50, 224, 224),
                                       (480, 854, 398, 224), (224, 224, 224, 224), (7, 9, 3, 5), (64, 64, 64, 31)])
def test_resize_bilinear(h, w, ow, oh):
    a = rnd(h * w, h, w, 3)
    ref = np.array(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    assert (I.resize_bilinear(a, (ow, oh)) == ref).all()


def test_gray_and_enhance():
    a = rnd(1, 64, 80, 3)
    im = Image.fromarray(a)
    assert (I.to_gray(a) == np.array(im.convert("L"))).all()
    for f in (0.0, 0.2, 0.5, 0.9999, 1.0, 1.3, 1.8):
        assert (I.enhance_brightness(a, f) == np.array(ImageEnhance.Brightness(im).enhance(f))).all()
        assert (I.enhance_contrast(a, f) == np.array(ImageEnhance.Contrast(im).enhance(f))).all()
        assert (I.enhance_saturation(a, f) == np.array(ImageEnhance.Color(im).enhance(f))).all()
    assert int(ImageStat.Stat(im.convert("L")).mean[0] + 0.5) == int(float(I.to_gray(a).astype(np.int64).sum()) / I.to_gray(a).size + 0.5)


def test_hsv_round_trip_pieces():
    a = rnd(2, 400, 400, 3)
    a[:20] = a[:20, :, :1]   # grays
    assert (I.rgb2hsv(a) == np.array(Image.fromarray(a).convert("HSV"))).all()
    hsv = rnd(3, 300, 300, 3)
    assert (I.hsv2rgb(hsv) == np.array(Image.fromarray(hsv, "HSV").convert("RGB"))).all()
    # torchvision's adjust_hue on the PIL backend (published algorithm): shift the H channel with uint8 wrap-around
    for hf in (-0.2, -0.07, 0.0, 0.13, 0.2):
        h, s, v = Image.fromarray(a).convert("HSV").split()
        nh = (np.array(h, dtype=np.int32) + int(hf * 255) % 256) % 256
        ref = np.array(Image.merge("HSV", (Image.fromarray(nh.astype(np.uint8), "L"), s, v)).convert("RGB"))
        assert (I.adjust_hue(a, hf) == ref).all()


@pytest.mark.parametrize("radius", [0.1, 0.37, 0.6123, 0.9, 1.3, 1.999, 2.0])
def test_gaussian_blur(radius):
    a = rnd(int(radius * 1000), 37, 53, 3)
    ref = np.array(Image.fromarray(a).filter(ImageFilter.GaussianBlur(radius=radius)))
    assert (I.gaussian_blur(a, radius) == ref).all()


def test_resized_crop_to_tensor():
    a = rnd(9, 90, 120, 3)
    crop = (5, 11, 60, 80)
    img = Image.fromarray(a).crop((11, 5, 11 + 80, 5 + 60)).resize((32, 32), Image.BILINEAR).transpose(Image.FLIP_LEFT_RIGHT)
    t = np.array(img).transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    mean, std = [0.485, 0.456, 0.406], [0.228, 0.224, 0.225]
    ref = (t - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    assert np.array_equal(I.resized_crop_to_tensor(a, crop, (32, 32), True, mean, std), ref)


def test_vectorised_tap_tables_match_the_loop_form():
    """The host module builds Pillow's tap tables with NumPy vector operations; they must equal the scalar loop restated in
    the oracle (which is the form pinned against Pillow above) for every size pair."""
    from timetuning_amd import video_transformations as VT

    rng = np.random.default_rng(0)
    pairs = list(zip(rng.integers(1, 1100, 300).tolist(), rng.integers(1, 600, 300).tolist())) + [(224, 224), (480, 224), (854, 398), (1, 5), (5, 1)]
    for a, b in pairs:
        k1, b1 = I.resample_coeffs(a, b)
        k2, b2 = VT.resample_coeffs(a, b)
        assert np.array_equal(k1, k2.numpy()) and np.array_equal(b1, b2.numpy()), (a, b)
