"""CPU oracle for the TimeTuning training hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch-CPU tensor code, the algorithm the reference runs for one
training iteration.  It is the checker the HIP path is
compared against; it is never imported by ``timetuning_amd`` (the product) and the only
callers allowed are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference itself from
``/root/reference`` (with stubs for its un-installed third-party imports), drives it on
seeded inputs and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function below against those vectors.  The reference has no tests or golden vectors of its
own, so these generated fixtures are the only pin that exists.

Each function cites the reference lines it follows (paths relative to ``/root/reference``).
The structure is deliberately the reference's own (two backbone passes per extractor call,
a Python loop over samples, fp64 label propagation on the host) because the same code is
the timed host-CPU baseline.
"""
from __future__ import annotations

import copy
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# ViT backbone  (dino_vision_transformer.py)
# --------------------------------------------------------------------------------------


def _ln(x, w, b, eps=1e-6):
    # nn.LayerNorm(eps=1e-6): dino_vision_transformer.py:283-287 (partial(nn.LayerNorm, eps=1e-6))
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def prepare_tokens(p, x, patch_size):
    """dino_vision_transformer.py:236-247 with PatchEmbed :166-171.  224x224 inputs take the
    identity branch of interpolate_pos_encoding (:214-218)."""
    B = x.shape[0]
    t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=patch_size)
    t = t.flatten(2).transpose(1, 2)
    cls = p["cls_token"].expand(B, -1, -1)
    t = torch.cat((cls, t), dim=1)
    return t + interpolate_pos_encoding(p["pos_embed"], t.shape[1] - 1, x.shape[2], x.shape[3], patch_size)


def interpolate_pos_encoding(pos_embed, npatch, w, h, patch_size):
    """dino_vision_transformer.py:214-234 (``w``/``h`` are the reference's names for the input's dims 2 and 3)."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    dim = pos_embed.shape[-1]
    w0, h0 = w // patch_size + 0.1, h // patch_size + 0.1
    g = int(math.sqrt(N))
    patch_pos = F.interpolate(pos_embed[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2), scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)),
                              mode="bicubic")
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    return torch.cat((pos_embed[:, 0].unsqueeze(0), patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def attention(p, pre, x, num_heads):
    """dino_vision_transformer.py:120-132.  Returns (projected output, attention probabilities)."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    y = F.linear(y, p[pre + "proj.weight"], p[pre + "proj.bias"])
    return y, attn


def block(p, i, x, num_heads, return_attention=False):
    """dino_vision_transformer.py:147-153 (drop_path is identity: drop rates are 0, :177-178)."""
    pre = f"blocks.{i}."
    y, attn = attention(p, pre + "attn.", _ln(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]), num_heads)
    if return_attention:
        return attn
    x = x + y
    h = _ln(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
    h = F.linear(h, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"])
    h = F.gelu(h)  # nn.GELU() = exact erf form (:89-105)
    h = F.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
    return x + h


def get_intermediate_layers(p, x, cfg):
    """dino_vision_transformer.py:265-273 with n=1: final-LayerNorm'd tokens of the last block."""
    t = prepare_tokens(p, x, cfg["patch_size"])
    for i in range(cfg["depth"]):
        t = block(p, i, t, cfg["num_heads"])
    return _ln(t, p["norm.weight"], p["norm.bias"])


def get_last_selfattention(p, x, cfg):
    """dino_vision_transformer.py:256-263: a second pass that returns block[-1]'s probabilities."""
    t = prepare_tokens(p, x, cfg["patch_size"])
    for i in range(cfg["depth"] - 1):
        t = block(p, i, t, cfg["num_heads"])
    return block(p, cfg["depth"] - 1, t, cfg["num_heads"], return_attention=True)


# --------------------------------------------------------------------------------------
# FeatureExtractor  (models.py:903-1078)
# --------------------------------------------------------------------------------------

SPATIAL_RESOLUTIONS = {"dino-s16": 14, "dino-b16": 14, "dino-s8": 28, "tiny-s16": 14}  # models.py:76


class FeatureExtractorOracle:
    """Functional twin of ``FeatureExtractor`` restricted to the "dino" branch (models.py:965-969)."""

    def __init__(self, cfg, backbone, head=None, unfreeze_layers=(), spatial_resolution=14):
        self.cfg = dict(cfg)
        self.backbone = OrderedDict(backbone)  # name -> tensor
        self.head = OrderedDict(head) if head else None
        self.spatial_resolution = spatial_resolution
        self.backbone_dim = cfg["embed_dim"]
        self.feature_dim = cfg["embed_dim"]
        if self.head:
            last = [k for k in self.head if k.endswith("weight")][-1]
            self.feature_dim = self.head[last].shape[0]
        # freeze_backbone (models.py:929-935): substring match on parameter names
        for name, t in self.backbone.items():
            t.requires_grad_(any(u in name for u in unfreeze_layers))
        if self.head:
            for t in self.head.values():
                t.requires_grad_(True)

    def named_parameters(self):
        for k, v in self.backbone.items():
            yield "backbone." + k, v
        if self.head:
            for k, v in self.head.items():
                yield "head." + k, v

    def parameters(self):
        return [v for _, v in self.named_parameters()]

    def clone(self):
        c = copy.copy(self)
        c.backbone = OrderedDict((k, v.detach().clone()) for k, v in self.backbone.items())
        c.head = OrderedDict((k, v.detach().clone()) for k, v in self.head.items()) if self.head else None
        return c

    def apply_head(self, x):
        # Linear GELU Linear GELU ... Linear (models.py:915-926)
        n_lin = len(self.head) // 2
        for i in range(n_lin):
            x = F.linear(x, self.head[f"{2 * i}.weight"], self.head[f"{2 * i}.bias"])
            if i != n_lin - 1:
                x = F.gelu(x)
        return x

    def get_features(self, x, faithful=True):
        feats = get_intermediate_layers(self.backbone, x, self.cfg)[:, 1:]
        attn = None
        if faithful:  # the reference always pays for the second pass (models.py:968)
            with torch.no_grad():
                attn = get_last_selfattention(self.backbone, x, self.cfg).detach()
        return feats, attn

    def forward(self, x, use_head=True, faithful=True):
        """models.py:1070-1078."""
        x, attn = self.get_features(x, faithful)
        if self.head is not None and use_head:
            ns, npatch, dim = x.shape
            x = self.apply_head(x.reshape(ns * npatch, dim)).view(ns, npatch, -1)
        return x, attn

    __call__ = forward


# --------------------------------------------------------------------------------------
# Sinkhorn-Knopp  (my_utils.py:246-274)  and schedules (my_utils.py:278-283)
# --------------------------------------------------------------------------------------


@torch.no_grad()
def sinkhorn(Q, nmb_iters, world_size=1, all_reduce=None):
    """my_utils.py:246-274.  ``Q`` is ``[K, B_local]``.  ``all_reduce`` is the SUM collective the
    reference issues through torch.distributed (:252,261,272); pass a callable for W>1."""
    Q = Q.detach().clone()
    sum_Q = torch.sum(Q)
    if world_size > 1:
        sum_Q = all_reduce(sum_Q)
    Q /= sum_Q
    K, B = Q.shape
    r = torch.ones(K, dtype=Q.dtype) / K
    c = torch.ones(B, dtype=Q.dtype) / (B * world_size)
    if world_size > 1:
        curr_sum = all_reduce(torch.sum(Q, dim=1))
    for _ in range(nmb_iters):
        u = curr_sum if world_size > 1 else torch.sum(Q, dim=1)
        Q *= (r / u).unsqueeze(1)
        Q *= (c / torch.sum(Q, dim=0)).unsqueeze(0)
        if world_size > 1:
            curr_sum = all_reduce(torch.sum(Q, dim=1))
    return (Q / torch.sum(Q, dim=0, keepdim=True)).t().float()


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep):
    """my_utils.py:278-283."""
    iters = np.arange(epochs * niter_per_ep)
    return final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))


# --------------------------------------------------------------------------------------
# Label propagation  (mask_propagation.py:377-496)
# --------------------------------------------------------------------------------------


def restrict_neighborhood(h, w, size_mask_neighborhood):
    """mask_propagation.py:377-391, vectorised: mask[(i,j),(i',j')] = |i-i'|<=r and |j-j'|<=r."""
    ii = torch.arange(h).view(h, 1, 1, 1)
    jj = torch.arange(w).view(1, w, 1, 1)
    pi = torch.arange(h).view(1, 1, h, 1)
    pj = torch.arange(w).view(1, 1, 1, w)
    m = ((ii - pi).abs() <= size_mask_neighborhood) & ((jj - pj).abs() <= size_mask_neighborhood)
    return m.reshape(h * w, h * w).float()


def label_propagation(size_mask_neighborhood, topk, spatial_resolution, frame_tar, list_frame_feats,
                      list_segs, mask_neighborhood):
    """mask_propagation.py:396-445 with ``features_exist=True``.

    frame_tar: [n, D] target-frame tokens; list_frame_feats: list of [D, n]; list_segs: list of
    [1, C, h, w] fp64.  Returns (seg_tar [1,C,h,w] fp64, frame_tar.T)."""
    h = w = spatial_resolution
    features = frame_tar
    return_feat_tar = features.T
    ncontext = len(list_frame_feats)
    feat_sources = torch.stack(list_frame_feats)                      # [c, D, n]
    feat_tar = F.normalize(features, dim=1, p=2)                      # over D
    feat_sources = F.normalize(feat_sources, dim=1, p=2)              # over D
    feat_tar = feat_tar.unsqueeze(0).repeat(ncontext, 1, 1)
    aff = torch.exp(torch.bmm(feat_tar, feat_sources) / 0.1)          # [c, n_tar, n_src]   (:422)
    if size_mask_neighborhood > 0:
        aff = aff * mask_neighborhood                                 # (:429)
    aff = aff.transpose(2, 1).reshape(-1, h * w)                      # [c*n_src, n_tar]    (:431)
    tk_val, _ = torch.topk(aff, dim=0, k=topk)
    tk_val_min, _ = torch.min(tk_val, dim=0)
    aff[aff < tk_val_min] = 0                                         # ties kept (:434)
    aff = aff / torch.sum(aff, keepdim=True, axis=0)                  # (:436)
    segs = torch.cat(list_segs)                                       # [c, C, h, w]
    nmb_context, C, h, w = segs.shape
    segs = segs.reshape(nmb_context, C, -1).transpose(2, 1).reshape(-1, C).T   # [C, c*n]   (:442)
    seg_tar = torch.mm(segs.double(), aff.double())                   # fp64 (:443)
    return seg_tar.reshape(1, C, h, w), return_feat_tar


def propagate_labels(n_last_frames, size_mask_neighborhood, topk, spatial_resolution, frame_list, first_seg):
    """mask_propagation.py:448-496 with ``features_exist=True``.

    frame_list: [fs, n, D]; first_seg: [1, C, g, g].  Returns the list of fs-1 maps [C,g,g] fp64."""
    # (:456) nearest-resize of the seed to the token grid: the identity on the training path, a real downsampling of
    # the first-frame annotation on the evaluation path
    first_seg = F.interpolate(first_seg.double(), size=(spatial_resolution, spatial_resolution), mode="nearest")
    mask = restrict_neighborhood(spatial_resolution, spatial_resolution, size_mask_neighborhood)
    frame1_feat = frame_list[0].T
    que = []
    out = []
    for cnt in range(1, frame_list.shape[0]):
        used_feats = [frame1_feat] + [pr[0] for pr in que]
        used_segs = [first_seg] + [pr[1] for pr in que]
        seg, feat_tar = label_propagation(size_mask_neighborhood, topk, spatial_resolution,
                                          frame_list[cnt], used_feats, used_segs, mask)
        if len(que) == n_last_frames:
            que.pop(0)
        que.append([feat_tar, seg])
        out.append(seg.squeeze(0))
    return out


def to_one_hot(y_tensor, n_dims=None):
    """mask_propagation.py:349-361: integer map [1,h,w] -> one-hot [n_dims,h,w]."""
    if n_dims is None:
        n_dims = int(y_tensor.max() + 1)
    _, h, w = y_tensor.size()
    y = y_tensor.long().view(-1, 1)
    return torch.zeros(y.size(0), n_dims).scatter_(1, y, 1).view(h, w, n_dims).permute(2, 0, 1)


def propagate_clip_predictions(n_last_frames, size_mask_neighborhood, topk, spatial_resolution, features, first_annotation,
                               input_resolution, n_classes=None, return_margin=False):
    """The per-clip body of the evaluation loop, mask_propagation.py:825-830: features [fs,n,D] (extractor without head),
    first_annotation [H,W] integer labels -> predictions [fs-1,R,R] (bilinear upsampling of the fp64 maps, arg-max)."""
    seed = to_one_hot(first_annotation.unsqueeze(0), n_classes).unsqueeze(0)
    maps = torch.stack(propagate_labels(n_last_frames, size_mask_neighborhood, topk, spatial_resolution, features, seed), dim=0)
    up = F.interpolate(maps, size=(input_resolution, input_resolution), mode="bilinear", align_corners=False)
    _, pred = torch.max(up, dim=1)
    if return_margin:
        top2 = up.topk(min(2, up.shape[1]), dim=1).values
        return pred, (top2[:, 0] - top2[:, -1]), maps
    return pred


def jaccard(pred, gt, num_classes, involve_bg=False):
    """Mean Jaccard index with identity label matching; classes absent from both maps are skipped, the background
    (class 0) is excluded unless ``involve_bg`` (PredsmIoU(..., involve_bg=False), mask_propagation.py:746)."""
    ious = []
    for c in range(0 if involve_bg else 1, num_classes):
        p, t = pred == c, gt == c
        union = (p | t).sum().item()
        if union:
            ious.append((p & t).sum().item() / union)
    return float(np.mean(ious)) if ious else float("nan")


# --------------------------------------------------------------------------------------
# evaluator: matched mIoU (metrics.py:246-505), proto_clustering (clustering.py:82-104), and NumPy restatements of the
# third-party pieces the evaluator delegates to (StandardScaler, faiss PCAMatrix, faiss Kmeans' Lloyd iteration) - the
# latter are "parity unpinned": faiss is not installed and its RNG stream / LAPACK sign choices are not reproducible.
# --------------------------------------------------------------------------------------


def miou(gt, pred, many_to_one=False, precision_based=False, involve_bg=False):
    """PredsmIoU.compute -> compute_miou (metrics.py:246-432) from flat integer label arrays.
    Returns (score, tp, fp, fn, reordered_preds, matched_bg_clusters)."""
    from scipy.optimize import linear_sum_assignment

    gt, pred = np.asarray(gt).astype(int), np.asarray(pred).astype(int)
    pred_unique, gt_unique = np.unique(pred), np.unique(gt)
    num_pred, num_gt = len(pred_unique), len(gt_unique)

    def score(c1, c2):  # get_score :435-455
        a, b = gt == c1, pred == c2
        tp_, fp_ = np.sum(a & b), np.sum(~a & b)
        if precision_based:
            return float(tp_) / max(float(tp_ + fp_), 1e-8)
        return float(tp_) / max(float(tp_ + fp_ + np.sum(a & ~b)), 1e-8)

    reordered = np.zeros(len(pred))
    if many_to_one:
        mat = np.array([score(c1, c2) for c2 in pred_unique for c1 in gt_unique]).reshape(num_pred, num_gt).T
        best = {}
        for pc in range(num_pred):
            for gc in range(num_gt):
                if pc not in best or mat[gc, pc] > best[pc][1]:
                    best[pc] = (gc, mat[gc, pc])
        for pc, (gc, _) in best.items():
            reordered[pred == pred_unique[pc]] = gt_unique[gc]
        bg = sum(1 for gc, _ in best.values() if gc == 0) / num_pred
    else:
        precision_based_saved, precision_based = precision_based, False   # _hungarian_match always matches on IoU (:476-483)
        mat = np.array([score(c1, c2) for c2 in pred_unique for c1 in gt_unique]).reshape(num_pred, num_gt).T
        precision_based = precision_based_saved
        rows, cols = linear_sum_assignment(1 - mat)
        for r, c in zip(rows, cols):
            reordered[pred == pred_unique[c]] = gt_unique[r]
        bg = 1 / num_gt
    tp, fp, fn, jac = {}, {}, {}, {}
    for g_ in gt_unique:
        a, b = gt == g_, reordered == g_
        tp[int(g_)], fp[int(g_)], fn[int(g_)] = int(np.sum(a & b)), int(np.sum(~a & b)), int(np.sum(a & ~b))
        jac[int(g_)] = float(tp[int(g_)]) / max(float(tp[int(g_)] + fp[int(g_)] + fn[int(g_)]), 1e-8)
    if not involve_bg:
        jac.pop(0, None)
        if len(jac) == 0:
            jac[0] = 0
    return float(np.mean(list(jac.values()))), tp, fp, fn, reordered.astype(int), bg


def proto_clustering(x, prototypes, input_size=14, output_size=224):
    """clustering.py:82-104 without the k-means merge of the prototypes (``num_classes=None``)."""
    n, num_patches, dim = x.shape
    xn, pn = F.normalize(x, dim=-1, p=2), F.normalize(prototypes, dim=-1, p=2)
    scores = torch.einsum("klm,nm->kln", xn, pn).permute(0, 2, 1).reshape(n, prototypes.shape[0], input_size, input_size)
    scores = F.interpolate(scores, size=(output_size, output_size), mode="bilinear", align_corners=False)
    return scores.permute(0, 2, 3, 1).argmax(dim=-1)


def standard_scale(x: np.ndarray) -> np.ndarray:
    """sklearn.preprocessing.StandardScaler as my_utils.py:23-30 drives it (partial_fit over row batches, then transform):
    per-column mean and POPULATION variance, scale = sqrt(var) with (near-)zero scales replaced by 1, in float64.  Pinned
    against the reference run with the real scikit-learn (tests/golden/scaler.npz)."""
    x = x.astype(np.float64)
    mean, std = x.mean(0), x.std(0)
    std[std < 10 * np.finfo(np.float64).eps] = 1.0
    return (x - mean) / std


def standard_scale_pca(x: np.ndarray, pca_dim: int):
    """StandardScaler (population variance, zero scales -> 1) followed by PCA onto the top ``pca_dim`` eigenvectors of the
    covariance of the standardised data (faiss.PCAMatrix with eigen_power 0); rows oriented so that their largest-magnitude
    entry is positive.  Returns (transformed [n, pca_dim], basis [pca_dim, dim])."""
    z = standard_scale(x)
    zc = z - z.mean(0)
    evals, evecs = np.linalg.eigh(zc.T @ zc / len(z))
    basis = evecs[:, np.argsort(evals)[::-1][:pca_dim]].T
    basis = basis * np.sign(basis[np.arange(len(basis)), np.abs(basis).argmax(1)])[:, None]
    return zc @ basis.T, basis


def kmeans_lloyd(x: np.ndarray, init_idx, niter: int):
    """Lloyd iterations from x[init_idx] (assignment to the nearest centroid, first minimum; means; empty clusters keep their
    centroid).  Returns (centroids, labels of the last assignment, objective of the last assignment)."""
    x = x.astype(np.float64)
    cent = x[np.asarray(init_idx)].copy()
    for _ in range(niter):
        d2 = ((x[:, None, :] - cent[None, :, :]) ** 2).sum(-1)
        lab = d2.argmin(1)
        obj = d2[np.arange(len(x)), lab].sum()
        for j in range(len(cent)):
            if (lab == j).any():
                cent[j] = x[lab == j].mean(0)
    return cent, lab, obj


# --------------------------------------------------------------------------------------
# attention foreground mask  (models.py:93-144, the --use_mask branch)
#
# PARITY OF THIS SECTION: the reference calls two third-party functions that are not installed
# in the build image and whose sources are not under /root/reference:
#   torchvision.transforms.GaussianBlur (torchvision==0.15.x per requirements) and
#   skimage.measure.label (scikit-image).
# ``gaussian_blur`` and ``label_components`` restate their PUBLISHED algorithms and are therefore
# "parity unpinned".  Everything around them (head mean, mass threshold, small-component removal,
# feature masking, masked cross-entropy) IS pinned: gen_golden.py runs the reference's own
# process_attentions / apply_attention_mask / get_loss with these two functions injected as the
# stand-ins for the missing imports.
# --------------------------------------------------------------------------------------


def gaussian_kernel1d(kernel_size: int, sigma: float) -> torch.Tensor:
    """torchvision.transforms.functional._get_gaussian_kernel1d: pdf on linspace(-h, h, k), normalised."""
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(img: torch.Tensor, kernel_size: int = 7, sigma: float = 0.6) -> torch.Tensor:
    """torchvision GaussianBlur(kernel_size, sigma) on [B,C,H,W]: reflect padding, depthwise conv with the outer
    product kernel.  GaussianBlur draws sigma from uniform_(s, s) in fp32, i.e. float32(sigma)."""
    sigma = float(torch.tensor(sigma, dtype=torch.float32))
    k1 = gaussian_kernel1d(kernel_size, sigma)
    k2 = torch.mm(k1[:, None], k1[None, :])
    C = img.shape[1]
    pad = kernel_size // 2
    x = F.pad(img, [pad, pad, pad, pad], mode="reflect")
    return F.conv2d(x, k2.expand(C, 1, kernel_size, kernel_size), groups=C)


def label_components(binary: np.ndarray) -> np.ndarray:
    """skimage.measure.label(arr) with its defaults: background 0, full connectivity (arr.ndim); here through
    scipy.ndimage.label with an all-ones structuring element.  Label numbering is not relied upon."""
    from scipy import ndimage

    lab, _ = ndimage.label(binary != 0, structure=np.ones((3,) * binary.ndim))
    return lab


def process_attentions(attentions: torch.Tensor, spatial_res: int, threshold: float = 0.65, blur_sigma: float = 0.6,
                       return_blurred: bool = False):
    """models.py:93-131.  attentions [F,h,N,N] -> {0,1} mask [F,1,g,g]."""
    attention = attentions[:, :, 0, 1:]
    bs, num_heads, _ = attention.shape
    attention = attention.reshape(bs, num_heads, spatial_res, spatial_res)
    attention = sum(attention[:, i] * 1 / num_heads for i in range(num_heads))  # :111-112
    attention = attention.reshape(bs, 1, spatial_res, spatial_res)
    attention = gaussian_blur(attention, 7, blur_sigma)
    blurred = attention.reshape(bs, spatial_res ** 2).clone()
    attention = attention.reshape(bs, 1, spatial_res ** 2)
    val, idx = torch.sort(attention)  # keep `threshold` of the mass (:117-123)
    val = val / torch.sum(val, dim=-1, keepdim=True)
    cumval = torch.cumsum(val, dim=-1)
    th_attn = cumval > (1 - threshold)
    idx2 = torch.argsort(idx)
    th_attn[:, 0] = torch.gather(th_attn[:, 0], dim=1, index=idx2[:, 0])
    th_attn = th_attn.reshape(bs, 1, spatial_res, spatial_res).float()
    for j in range(bs):  # components of <= 2 pixels are dropped (:125-130)
        labelled = label_components(th_attn[j].numpy())
        for k in range(1, int(labelled.max()) + 1):
            comp = labelled == k
            if comp.sum() <= 2:
                th_attn[j, 0][torch.from_numpy(comp[0])] = 0
    th_attn = th_attn.detach()
    if return_blurred:
        # distance of the sorted cumulative mass to the cut, per pixel: tests use it to excuse pixels whose side of the
        # threshold depends on the last bits of the attention values
        margin = torch.gather((cumval - (1 - threshold)).abs()[:, 0], 1, idx2[:, 0])
        return th_attn, blurred, margin
    return th_attn


def apply_attention_mask(features: torch.Tensor, attentions: torch.Tensor, spatial_resolution: int):
    """models.py:133-144.  features [bs,fs,n,dim], attentions [bs*fs,h,N,N] -> (masked features, mask [bs,fs,n])."""
    mask = process_attentions(attentions, spatial_resolution)
    bs, fs, n, dim = features.shape
    mask = mask.view(bs, fs, n, 1)
    return features * mask, mask.squeeze()


# --------------------------------------------------------------------------------------
# TimeT objective  (time_tuning.py:80-302)
# --------------------------------------------------------------------------------------


class TimeTOracle:
    def __init__(self, feature_extractor: FeatureExtractorOracle, prototypes: torch.Tensor, world_size=1,
                 all_reduce=None):
        self.feature_extractor = feature_extractor
        self.prototypes = prototypes.clone().requires_grad_(True)
        self.teacher = None
        self.teacher_prototypes = None
        self.queue = None
        self.momentum_schedule = None
        self.world_size = world_size
        self.all_reduce = all_reduce

    # -- parameters in the reference's named_parameters() order (prototypes first: :93) ------
    def named_parameters(self):
        yield "prototypes", self.prototypes
        for k, v in self.feature_extractor.named_parameters():
            yield "feature_extractor." + k, v

    def init_momentum_teacher(self):
        """time_tuning.py:96-104."""
        self.teacher = self.feature_extractor.clone()
        for t in self.teacher.parameters():
            t.requires_grad_(False)
        self.teacher_prototypes = self.prototypes.detach().clone()

    def init_queue(self, queue_size):
        """time_tuning.py:106-107."""
        self.queue = torch.zeros((queue_size, self.feature_extractor.feature_dim), dtype=self.prototypes.dtype)

    def set_momentum_teacher_schedular_params(self, m0, m1, epochs, iters):
        self.momentum_schedule = cosine_scheduler(m0, m1, epochs, iters)

    @torch.no_grad()
    def update_momentum_teacher(self, step):
        """time_tuning.py:109-118: teacher <- teacher*(1-m) + student*m, m = schedule[step]."""
        m = self.momentum_schedule[step]
        for q, k in zip(self.feature_extractor.parameters(), self.teacher.parameters()):
            k.copy_(k * (1.0 - m) + q.detach() * m)
        tp = self.teacher_prototypes * (1.0 - m) + self.prototypes.detach() * m
        self.teacher_prototypes = F.normalize(tp, dim=1, p=2)

    @torch.no_grad()
    def normalize_prototypes(self):
        """time_tuning.py:124-128."""
        self.prototypes.copy_(F.normalize(self.prototypes.detach().clone(), dim=1, p=2))

    def get_feature_prototype_similarity(self, x, use_teacher=False):
        """time_tuning.py:130-141."""
        nx = F.normalize(x, dim=-1, p=2)
        protos = self.teacher_prototypes if use_teacher else self.prototypes
        return torch.mm(nx, protos.t())

    def find_optimal_assignment(self, scores, epsilon, sinkhorn_iterations):
        """time_tuning.py:157-168."""
        with torch.no_grad():
            q = torch.exp(scores / epsilon).t()
            return sinkhorn(q, sinkhorn_iterations, self.world_size, self.all_reduce)

    def get_scores(self, features, epsilon, sinkhorn_iterations, use_teacher=False):
        """time_tuning.py:195-217."""
        bs, npatch, dim = features.shape
        flat = features.contiguous().view(bs * npatch, dim)
        batch_scores = self.get_feature_prototype_similarity(flat, use_teacher)
        scores = batch_scores
        if self.queue is not None and self.queue[-1].count_nonzero() != 0:
            queue_scores = self.get_feature_prototype_similarity(self.queue.view(-1, dim), use_teacher)
            scores = torch.cat([batch_scores, queue_scores], dim=0)
        q = self.find_optimal_assignment(scores, epsilon, sinkhorn_iterations)
        return q[: bs * npatch].view(bs, npatch, -1), batch_scores.view(bs, npatch, -1)

    def make_seg_maps(self, q_i, feats_i, n_last_frames, size_mask_neighborhood, topk):
        """time_tuning.py:143-154."""
        g = self.feature_extractor.spatial_resolution
        seed = q_i.view(g, g, -1).permute(2, 0, 1).unsqueeze(0)
        return torch.stack(propagate_labels(n_last_frames, size_mask_neighborhood, topk, g, feats_i, seed))

    def get_loss(self, x, n_last_frames=7, size_mask_neighborhood=6, topk=5, epsilon=0.05,
                 sinkhorn_iterations=10, queue_perm=None, faithful=True, return_aux=False, mask_features=False,
                 labels_override=None, masks_override=None):
        """time_tuning.py:224-302; ``mask_features`` is the --use_mask branch (:226-227,235-236,244-246,282-283,298-299).

        ``queue_perm``: the permutation ``torch.randperm(bs*n)`` draws at :259; passing it makes the
        queue update reproducible.  ``faithful`` keeps the reference's redundant passes.
        Checker aids (the two discontinuous decisions of the objective, pinned so that everything continuous around them can be
        compared unconditionally): ``labels_override`` [bs, g*g] replaces the arg-max labels in the cross entropy;
        ``masks_override = (source_mask [bs, n], target_mask [bs, n])`` replaces the foreground masks of the source frame
        (the teacher's when there is one) and of the target frame."""
        fe = self.feature_extractor
        g = fe.spatial_resolution
        bs, fs, c, h, w = x.shape
        flat = x.view(bs * fs, c, h, w)
        faithful = faithful or mask_features  # the masks need the last block's attention
        if self.teacher is not None:
            with torch.no_grad():
                teacher_features, teacher_attentions = self.teacher(flat, faithful=faithful)
            teacher_features = teacher_features.view(bs, fs, *teacher_features.shape[1:])
            if mask_features:
                if masks_override is not None:
                    tm = process_attentions(teacher_attentions, g).view(bs, fs, -1).clone()
                    tm[:, 0] = torch.as_tensor(masks_override[0]).float().view(bs, -1)
                    teacher_features = teacher_features * tm.unsqueeze(-1)
                else:
                    teacher_features, _ = apply_attention_mask(teacher_features, teacher_attentions, g)
        features, attentions = fe(flat, faithful=faithful)
        with torch.no_grad():
            if faithful:
                backbone_features, _ = fe(flat, use_head=False, faithful=True)
            else:
                backbone_features, _ = fe(flat, use_head=False, faithful=False)
        npatch, dim = features.shape[1:]
        features = features.view(bs, fs, npatch, dim)
        backbone_features = backbone_features.view(bs, fs, npatch, -1)
        masks = None
        if mask_features:
            if masks_override is not None:
                masks = process_attentions(attentions, g).view(bs, fs, -1).clone()
                if self.teacher is None:
                    masks[:, 0] = torch.as_tensor(masks_override[0]).float().view(bs, -1)
                masks[:, -1] = torch.as_tensor(masks_override[1]).float().view(bs, -1)
                features = features * masks.unsqueeze(-1)
            else:
                features, masks = apply_attention_mask(features, attentions, g)
            masks = masks.view(bs, fs, g, g)
        source_features = features[:, 0]

        if self.queue is not None:  # :250-261
            qf = (teacher_features[:, 0] if self.teacher is not None else features[:, 0]).reshape(-1, dim)
            m = min(bs * 10, self.queue.size(0))
            perm = torch.randperm(qf.size(0)) if queue_perm is None else torch.as_tensor(queue_perm)
            idx = perm[:m]
            self.queue[m:] = self.queue[:-m].clone()
            self.queue[:m] = qf[idx].detach()

        if self.teacher is not None:  # :263-275
            batch_q = self.get_scores(teacher_features[:, 0], epsilon, sinkhorn_iterations, use_teacher=True)[0]
            if faithful:
                self.get_scores(source_features, epsilon, sinkhorn_iterations)
                self.get_scores(teacher_features[:, -1], epsilon, sinkhorn_iterations, use_teacher=True)
            target_batch_scores = self.get_scores(features[:, -1], epsilon, sinkhorn_iterations)[1]
        else:
            batch_q, _ = self.get_scores(source_features, epsilon, sinkhorn_iterations)
            _, target_batch_scores = self.get_scores(features[:, -1], epsilon, sinkhorn_iterations)

        batch_loss = 0
        labels_all, pmaps = [], []
        for i in range(bs):  # :277-301
            maps = self.make_seg_maps(batch_q[i], backbone_features[i], n_last_frames, size_mask_neighborhood, topk)
            p_map = maps[-1]
            target_scores = target_batch_scores[i].view(g, g, -1).permute(2, 0, 1)
            labels = p_map.unsqueeze(0).argmax(dim=1).long()
            labels_all.append(labels[0])
            if labels_override is not None:
                labels = torch.as_tensor(labels_override)[i].long().view(1, g, g)
            if mask_features:  # reduction='none', weighted by the target frame's mask, mean over ALL g*g patches
                loss = F.cross_entropy(target_scores.unsqueeze(0) / 0.1, labels, reduction="none") * masks[i, -1].unsqueeze(0)
            else:
                loss = F.cross_entropy(target_scores.unsqueeze(0) / 0.1, labels)
            batch_loss = batch_loss + loss.mean()
            pmaps.append(p_map)
        loss = batch_loss / bs
        if return_aux:
            return loss, dict(batch_q=batch_q, target_scores=target_batch_scores, labels=torch.stack(labels_all),
                              p_map=torch.stack(pmaps), features=features, backbone_features=backbone_features, masks=masks)
        return loss


# --------------------------------------------------------------------------------------
# SwavOptimizer  (time_tuning.py:379-429): AdamW + cosine LR + weight-decay reschedule
# --------------------------------------------------------------------------------------


def param_groups(model: TimeTOracle, backbone_lr, lr, weight_decay):
    """time_tuning.py:391-415: groups ordered prototypes, head, backbone; each split into
    (decayed, undecayed[bias or 1-D])."""
    groups = []
    for filt, g_lr in (("prototypes", lr), ("feature_extractor.head", lr), ("feature_extractor.backbone", backbone_lr)):
        dec, nodec = [], []
        for name, p in model.named_parameters():
            if p.requires_grad and filt in name:
                (nodec if (name.endswith(".bias") or p.dim() == 1) else dec).append((name, p))
        groups.append(dict(params=dec, weight_decay=weight_decay, lr=g_lr, base_lr=g_lr))
        groups.append(dict(params=nodec, weight_decay=0.0, lr=g_lr, base_lr=g_lr))
    return groups


class SwavOptimizerOracle:
    """Restates torch.optim.AdamW (betas (0.9, 0.999), eps 1e-8, decoupled decay) as driven by
    time_tuning.py:420-429, plus CosineAnnealingLR(T_max=I*E, eta_min=0)."""

    def __init__(self, model, backbone_lr, lr, wd_schedule, num_itr, num_epochs, use_lr_scheduler=True):
        self.groups = param_groups(model, backbone_lr, lr, wd_schedule[0])
        self.wd_schedule = wd_schedule
        self.T_max = num_itr * num_epochs
        self.use_lr_scheduler = use_lr_scheduler
        self.global_step = 0
        self.state = {}

    @torch.no_grad()
    def step(self):
        b1, b2, eps = 0.9, 0.999, 1e-8
        for g in self.groups:
            for name, p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.setdefault(name, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
                st["step"] += 1
                t = st["step"]
                p.mul_(1 - g["lr"] * g["weight_decay"])
                st["m"].mul_(b1).add_(p.grad, alpha=1 - b1)
                st["v"].mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
                bc1 = 1 - b1 ** t
                bc2 = 1 - b2 ** t
                denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(st["m"], denom, value=-g["lr"] / bc1)
        self.global_step += 1
        if self.use_lr_scheduler:  # closed form of CosineAnnealingLR with eta_min = 0
            for g in self.groups:
                g["lr"] = g["base_lr"] * 0.5 * (1 + math.cos(math.pi * self.global_step / self.T_max))
        for g in self.groups:  # :427-429 (IndexError on the very last step is the reference's own)
            if g["weight_decay"] != 0:
                g["weight_decay"] = float(self.wd_schedule[self.global_step])

    def zero_grad(self):
        for g in self.groups:
            for _, p in g["params"]:
                p.grad = None


# --------------------------------------------------------------------------------------
# builders
# --------------------------------------------------------------------------------------


def build_oracle(arch="dino-s16", num_prototypes=200, head_layer_list=(1024, 1024, 512, 256), mode="dino",
                 seed=1, dtype=torch.float32, unfreeze_layers=("blocks.11", "blocks.10"), vit_cfg=None,
                 world_size=1, all_reduce=None):
    """Model with the portable synthetic weights (timetuning_amd.synth)."""
    from timetuning_amd import synth

    cfg = dict(vit_cfg or synth.ARCHS[arch])
    bb = OrderedDict((k, torch.from_numpy(v).to(dtype)) for k, v in synth.make_vit_weights(mode=mode, seed=seed, **cfg).items())
    head = None
    if head_layer_list:
        head = OrderedDict((k, torch.from_numpy(v).to(dtype))
                           for k, v in synth.make_head_weights(cfg["embed_dim"], head_layer_list, mode=mode, seed=seed).items())
    fe = FeatureExtractorOracle(cfg, bb, head, unfreeze_layers, SPATIAL_RESOLUTIONS.get(arch, 224 // cfg["patch_size"]))
    protos = torch.from_numpy(synth.make_prototypes(num_prototypes, fe.feature_dim, seed)).to(dtype)
    return TimeTOracle(fe, protos, world_size, all_reduce)
