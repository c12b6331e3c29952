"""Label propagation with the reference's surface (``mask_propagation.py``) on the HIP kernels.

``propagate_labels`` / ``to_one_hot`` keep the reference signatures (``mask_propagation.py:349-361,448-496``);
``propagate_clip`` is the per-clip body of the evaluation loop (``:821-831``: extractor without head ->
``propagate_labels`` -> bilinear upsampling -> arg-max) and ``jaccard`` scores the propagated masks.  The evaluation
driver reproduces the reference's flag set (``:849-871``, DAVIS protocol defaults ``--n_last_frames 4
--size_mask_neighborhood 12 --topk 5``); dataset readers are out of scope, so it runs on synthetic clips.
"""
from __future__ import annotations

import argparse
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import hip_ops as ops


def to_one_hot(y_tensor: torch.Tensor, n_dims: Optional[int] = None) -> torch.Tensor:
    """Integer map [1,h,w] -> one-hot [n_dims,h,w] float (``mask_propagation.py:349-361``)."""
    if n_dims is None:
        n_dims = int(y_tensor.max() + 1)
    _, h, w = y_tensor.size()
    idx = y_tensor.long().reshape(-1, 1)
    one_hot = torch.zeros(idx.shape[0], n_dims, device=y_tensor.device).scatter_(1, idx, 1)
    return one_hot.view(h, w, n_dims).permute(2, 0, 1)


def _maps(n_last_frames, size_mask_neighborhood, topk, model, frame_list, first_seg, features_exist):
    fe = model.feature_extractor if hasattr(model, "feature_extractor") else model
    g = fe.spatial_resolution
    if not features_exist:
        frame_list, _ = fe(frame_list, use_head=False)
    fs, n, D = frame_list.shape
    # the seed is resized to the token grid with nearest-neighbour sampling, in fp64 as the reference (:456)
    first_seg = F.interpolate(first_seg.double(), size=(g, g), mode="nearest")
    C = first_seg.shape[1]
    xn = ops.l2norm_fwd(frame_list.reshape(fs * n, D).contiguous().float()).view(fs, 1, n, D)
    seed = first_seg.reshape(C, n).t().contiguous().float().view(1, n, C).to(xn.device)
    return ops.label_propagate_maps(xn, seed, n_last_frames, size_mask_neighborhood, topk, 0.1), C, g  # [fs-1, 1, n, C]


@torch.no_grad()
def propagate_labels(n_last_frames, size_mask_neighborhood, topk, model, frame_list, first_seg, features_exist=False) -> List[torch.Tensor]:
    """frame_list [fs, n, D] backbone tokens (``features_exist=True``) or [fs, 3, H, W] frames; first_seg [1, C, h, w].
    Returns the fs-1 propagated maps ``[C, g, g]`` fp64, as the reference (``mask_propagation.py:448-496``)."""
    maps, C, g = _maps(n_last_frames, size_mask_neighborhood, topk, model, frame_list, first_seg, features_exist)
    return [m[0].t().reshape(C, g, g) for m in maps]


@torch.no_grad()
def propagate_clip(model, clip: torch.Tensor, first_annotation: torch.Tensor, n_last_frames: int = 4, size_mask_neighborhood: int = 12,
                   topk: int = 5, input_resolution: int = 224, num_classes: Optional[int] = None) -> torch.Tensor:
    """One clip of the evaluation loop (``mask_propagation.py:824-830``): clip [fs,3,H,W], first_annotation [H,W] integer
    labels of frame 0 -> predictions [fs-1, R, R] int64 for frames 1..fs-1."""
    fe = model.feature_extractor if hasattr(model, "feature_extractor") else model
    feats, _ = fe(clip, use_head=False)
    seed = to_one_hot(first_annotation.unsqueeze(0), num_classes).unsqueeze(0)
    maps, C, g = _maps(n_last_frames, size_mask_neighborhood, topk, model, feats, seed, True)
    return ops.upsample_argmax(maps.view(maps.shape[0], g * g, C), input_resolution)


@torch.no_grad()
def jaccard(pred: torch.Tensor, gt: torch.Tensor, num_classes: int, involve_bg: bool = False) -> Tuple[float, torch.Tensor]:
    """Mean Jaccard index (J) of integer label maps with IDENTITY label matching (propagated labels keep their ids) and the
    per-class values.  Classes absent from both prediction and ground truth are skipped; the background (class 0) is
    excluded unless ``involve_bg`` (the reference builds ``PredsmIoU(num_clusters, 10, involve_bg=False)``, :746).  The
    reference's Hungarian / many-to-one matching of ``evaluate_localizations`` belongs to the clustering evaluator and
    is not part of this build."""
    counts = ops.confusion_counts(pred.contiguous().view(-1), gt.contiguous().view(-1).long(), num_classes).double()
    inter = counts.diagonal()
    union = counts.sum(0) + counts.sum(1) - inter
    valid = union > 0
    if not involve_bg:
        valid[0] = False
    iou = torch.where(valid, inter / union.clamp(min=1), torch.full_like(inter, float("nan")))
    return (float(iou[valid].mean()) if valid.any() else float("nan")), iou


def build_parser() -> argparse.ArgumentParser:
    """Flag names and defaults of ``mask_propagation.py:849-871`` (``type=bool`` flags keep the any-non-empty-string-is-True
    quirk).  ``--dataset synthetic`` and ``--num_clips`` are additions: the dataset readers are out of scope."""
    p = argparse.ArgumentParser()
    p.add_argument("--architecture", type=str, default="dino-s16")
    p.add_argument("--model_path", type=str, default="../models/leopart_vits16.ckpt")
    p.add_argument("--dataset", type=str, default="davis_val")
    p.add_argument("--dataset_path", type=str, default="../data")
    p.add_argument("--destination_path", type=str, default="ytvos")
    p.add_argument("--evaluation_protocol", type=str, default="frame-wise")
    p.add_argument("--logging_directory", type=str, default="visualizations")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--num_workers", type=int, default=10)
    p.add_argument("--num_clusters", type=int, default=10)
    p.add_argument("--input_resolution", type=int, default=224)
    p.add_argument("--many_to_one", type=bool, default=False)
    p.add_argument("--num_frames", type=int, default=25)
    p.add_argument("--n_last_frames", type=int, default=4)
    p.add_argument("--uvos", type=int, default=True)
    p.add_argument("--topk", type=int, default=5)
    p.add_argument("--size_mask_neighborhood", default=12, type=int)
    p.add_argument("--epsilon", default=0.05, type=float)
    p.add_argument("--sinkhorn_iterations", default=3, type=float)
    p.add_argument("--use_projection_head", type=bool, default=True)
    p.add_argument("--use_optical_flow", type=bool, default=False)
    p.add_argument("--num_clips", type=int, default=4, help="synthetic data only")
    return p


def synthetic_tracking_clip(fs: int, resolution: int, seed: int, objects: int = 2):
    """A clip with ``objects`` textured discs drifting over a textured background, and its per-frame integer masks:
    frames [fs,3,R,R] fp32 (roughly unit-normal, like normalised images), masks [fs,R,R] int64 (0 = background)."""
    import numpy as np

    from . import synth

    R = resolution
    yy, xx = np.mgrid[0:R, 0:R].astype(np.float32)
    tex = synth.normal("trk.tex", (objects + 1, 3, 8, 8), 1.0, 0.0, seed)
    tex = np.kron(tex, np.ones((1, 1, R // 8, R // 8), np.float32))[:, :, :R, :R]
    pos = synth.normal("trk.pos", (objects, 4), 1.0, 0.0, seed)
    frames, masks = [], []
    for t in range(fs):
        img = tex[0].copy()
        m = np.zeros((R, R), np.int64)
        for o in range(objects):
            cy = R * (0.3 + 0.4 * o / max(objects - 1, 1)) + 3.0 * t * np.tanh(pos[o, 0])
            cx = R * (0.3 + 0.2 * o) + 4.0 * t * np.tanh(pos[o, 1])
            rad = R * (0.12 + 0.03 * abs(np.tanh(pos[o, 2])))
            inside = (yy - cy) ** 2 + (xx - cx) ** 2 < rad ** 2
            img = np.where(inside[None], tex[o + 1] + 1.5 * (o + 1), img)
            m[inside] = o + 1
        frames.append(img + 0.05 * synth.normal(f"trk.noise.{t}", (3, R, R), 1.0, 0.0, seed))
        masks.append(m)
    return torch.from_numpy(np.stack(frames).astype(np.float32)), torch.from_numpy(np.stack(masks))


def mask_propagation(args) -> float:
    """The evaluation loop of ``mask_propagation.py:757-846`` on synthetic clips; returns the mean J over clips."""
    from .models import FeatureExtractor
    from .time_tuning import TimeT

    if args.dataset != "synthetic":
        raise NotImplementedError("dataset readers (data_loader.py) are out of scope for this build; run with --dataset synthetic")
    if args.use_optical_flow:
        raise NotImplementedError("the optical-flow baseline (cv2 Farneback, mask_propagation.py:803-815) is not part of this build")
    device = torch.device("cuda", 0)
    fe = FeatureExtractor(args.architecture, args.model_path, [1024, 1024, 512, 256], return_attention=False)  # "" = synthetic weights
    model = TimeT(fe, 200).to(device).eval()
    scores = []
    for i in range(args.num_clips):
        clip, masks = synthetic_tracking_clip(args.num_frames, args.input_resolution, seed=i + 1)
        if args.uvos:  # all objects become one foreground class (:797-799)
            masks = (masks > 0).long()
        C = int(masks.max()) + 1
        pred = propagate_clip(model, clip.to(device), masks[0].to(device), args.n_last_frames, args.size_mask_neighborhood, args.topk,
                              args.input_resolution, C)
        j, _ = jaccard(pred, masks[1:].to(device), C)
        scores.append(j)
        print(f"clip {i}: J = {j:.4f}")
    mean = sum(scores) / len(scores)
    print(f"mean J over {len(scores)} clips: {mean:.4f}")
    return mean


if __name__ == "__main__":
    mask_propagation(build_parser().parse_args())
