"""GPU parity of the clip input pipeline: every kernel bit for bit against the Pillow-pinned oracle
(oracle/image_ops.py), and the reference-named transform classes against outputs of the reference's own Compose pipelines
(tests/golden/transforms.npz) with the same seeds - the random decisions and the pixels must both agree."""
import random

import numpy as np
import pytest
import torch

from oracle import image_ops as I

pytestmark = pytest.mark.gpu
MEAN, STD = [0.485, 0.456, 0.406], [0.228, 0.224, 0.225]


def rnd(seed, *shape):
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def dev(a):
    return torch.as_tensor(a).cuda().contiguous()


@pytest.mark.parametrize("h,w,oh,ow", [(100, 160, 224, 224), (300, 500, 224, 224), (333, 257, 298, 224), (50, 50, 224, 224), (7, 9, 5, 3),
                                       (64, 64, 31, 64), (64, 64, 64, 31), (40, 40, 40, 40)])
def test_resize_matches_pillow_oracle(h, w, oh, ow):
    from timetuning_amd import video_transformations as VT

    a = rnd(h * w + oh, 2, h, w, 3)
    got = VT.resized_crop(dev(a), 0, 0, h, w, (oh, ow)).cpu().numpy()
    for f in range(2):
        assert (got[f] == I.resize_bilinear(a[f], (ow, oh))).all()


def test_resized_crop_to_tensor_and_flip():
    from timetuning_amd import video_transformations as VT

    a = rnd(5, 3, 90, 120, 3)
    for crop, size, flip in (((5, 11, 60, 80), (32, 32), False), ((0, 0, 90, 120), (64, 48), True), ((10, 0, 33, 120), (33, 120), True),
                             ((0, 7, 90, 50), (90, 50), False)):
        i, j, h, w = crop
        got = VT.resized_crop(dev(a), i, j, h, w, size, to_tensor=(MEAN, STD), flip=flip).cpu().numpy()
        for f in range(3):
            assert np.array_equal(got[f], I.resized_crop_to_tensor(a[f], crop, size, flip, MEAN, STD)), (crop, size, flip)
        u8 = VT.resized_crop(dev(a), i, j, h, w, size).cpu().numpy()
        assert (u8[0] == I.resize_bilinear(a[0][i:i + h, j:j + w], (size[1], size[0]))).all()


def test_colour_ops():
    from timetuning_amd import hip_ops as ops

    a = rnd(7, 2, 80, 96, 3)
    a[0, :10] = a[0, :10, :, :1]  # some exact grays for the HSV path
    g = ops.img_color_(dev(a), ops.IMG_GRAYSCALE).cpu().numpy()
    assert (g == np.stack([I.gray3(x) for x in a])).all()
    for f in (0.0, 0.2, 0.5, 0.9999, 1.0, 1.3, 1.8):
        assert (ops.img_color_(dev(a), ops.IMG_BRIGHTNESS, f).cpu().numpy() == np.stack([I.enhance_brightness(x, f) for x in a])).all()
        assert (ops.img_color_(dev(a), ops.IMG_CONTRAST, f).cpu().numpy() == np.stack([I.enhance_contrast(x, f) for x in a])).all()
        assert (ops.img_color_(dev(a), ops.IMG_SATURATION, f).cpu().numpy() == np.stack([I.enhance_saturation(x, f) for x in a])).all()
    for hf in (-0.2, -0.07, 0.0, 0.13, 0.2):
        got = ops.img_color_(dev(a), ops.IMG_HUE, 1.0, I.hue_shift_u8(hf)).cpu().numpy()
        assert (got == np.stack([I.adjust_hue(x, hf) for x in a])).all(), hf
    big = rnd(8, 1, 512, 512, 3)   # 262144 random colours through rgb -> hsv -> rgb
    assert (ops.img_color_(dev(big), ops.IMG_HUE, 1.0, 37).cpu().numpy()[0] == I.adjust_hue(big[0], 37 / 255 + 1e-9)).all()


@pytest.mark.parametrize("radius", [0.1, 0.37, 0.6123, 0.9, 1.3, 1.999, 2.0])
def test_gaussian_blur(radius):
    from timetuning_amd import video_transformations as VT

    a = rnd(int(radius * 1000), 2, 37, 53, 3)
    got = VT.gaussian_blur(dev(a), radius).cpu().numpy()
    for f in range(2):
        assert (got[f] == I.gaussian_blur(a[f], radius)).all()


@pytest.mark.parametrize("tag,seeds", [("a", range(7)), ("b", range(3))])
def test_training_transforms_match_reference(golden, tag, seeds):
    """frame_transform (colour jitter / grayscale / blur) then video_transform (Resize -> RandomResizedCrop -> flip ->
    ClipToTensor) as time_tuning.py:588-593 builds them, seeded like the reference run that produced the fixture."""
    from timetuning_amd import video_transformations as VT

    d = golden("transforms")
    frames = dev(d[f"{tag}_frames"])
    data_transform, video_transform = VT.training_transforms(64)
    branches = set()
    for seed in seeds:
        random.seed(seed)
        torch.manual_seed(seed)
        clip = data_transform(frames)
        key = f"{tag}_seed{seed}_after_frame_transform"
        if key in d.files:
            assert (clip.cpu().numpy() == d[key]).all(), seed
            branches.add("changed" if (d[key] != d[f"{tag}_frames"]).any() else "same")
        out = video_transform(clip)
        assert out.shape == (3, 3, 64, 64) and out.dtype == torch.float32
        assert np.array_equal(out.cpu().numpy(), d[f"{tag}_seed{seed}"]), seed
    if tag == "a":
        assert branches == {"changed", "same"}


def test_cli_driver_on_raw_frames(tmp_path):
    """python -m timetuning_amd.time_tuning --dataset synthetic_frames: uint8 frames -> GPU transforms -> TimeT steps with the
    reference's flags (teacher, queue and --use_mask on: every branch of the step in one run)."""
    from timetuning_amd.time_tuning import build_parser, time_tuning

    args = build_parser().parse_args(["--dataset", "synthetic_frames", "--model_path", "", "--batch_size", "2", "--num_frames", "2",
                                      "--num_clusters", "20", "--num_epochs", "1", "--steps_per_epoch", "2", "--use_queue", "1",
                                      "--queue_size", "128", "--use_mask", "1", "--logging_directory", str(tmp_path)])
    model = time_tuning(0, args)
    assert model.teacher is not None and model.queue is not None
    assert torch.isfinite(model.prototypes).all() and (tmp_path / "checkpoint.pth").exists()
    # the periodic rank-0 evaluation (time_tuning.py:634-646) ran at epoch 0 on the synthetic evaluation set: k-means over
    # backbone features -> matched mIoU; the best-score snapshot is written like the reference's "{score}_{epoch}.pth"
    assert len(model.eval_scores) == 1 and model.eval_scores[0][0] == 0 and 0.0 < model.eval_scores[0][1] <= 1.0
    assert (tmp_path / f"{model.eval_scores[0][1]}_0.pth").exists()
