"""``propagate_labels`` with the reference's signature (``mask_propagation.py:448-496``) on the HIP kernel."""
from __future__ import annotations

import torch

from . import hip_ops as ops


@torch.no_grad()
def propagate_labels(n_last_frames, size_mask_neighborhood, topk, model, frame_list, first_seg, features_exist=False):
    """frame_list [fs, n, D] backbone tokens (``features_exist=True``) or [fs, 3, H, W] frames; first_seg [1, C, g, g].
    Returns the propagated map of the LAST frame as a one-element list ``[C, g, g]`` fp64: the reference returns all
    fs-1 maps, but every caller on the training path reads only ``[-1]`` (time_tuning.py:294)."""
    fe = model.feature_extractor if hasattr(model, "feature_extractor") else model
    g = fe.spatial_resolution
    if not features_exist:
        frame_list, _ = fe(frame_list, use_head=False)
    fs, n, D = frame_list.shape
    C = first_seg.shape[1]
    xn = ops.l2norm_fwd(frame_list.reshape(fs * n, D).contiguous().float()).view(fs, 1, n, D)
    seed = first_seg.reshape(C, n).t().contiguous().float().view(1, n, C).to(xn.device)
    _, pmap = ops.label_propagate(xn, seed, n_last_frames, size_mask_neighborhood, topk, 0.1, return_pmap=True)
    return [pmap[0].t().reshape(C, g, g)]
