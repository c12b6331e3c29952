"""Evaluator clustering with the reference's surface (``clustering.py:20-117``, ``my_utils.py:19-37``) on the HIP kernels.

``cluster_features`` / ``proto_clustering`` / ``normalize_and_transform`` keep the reference's names, arguments and output
layouts.  The reference leans on two third-party CPU libraries here: scikit-learn's ``StandardScaler`` and faiss
(``PCAMatrix``, ``Kmeans``).  Their published algorithms are restated on the GPU:

* StandardScaler        per-column mean / population variance (``tt_col_moments``), scale = sqrt(var) with zeros -> 1
                        (pinned: tests/golden/scaler.npz is the reference's run with the real scikit-learn)
* faiss.PCAMatrix(d, p) eigenvectors of the covariance of the (standardised) data, largest eigenvalues first; the p x d basis
                        comes from a d x d eigen-problem solved on the host in fp64, the Gram matrix and the projection are
                        device GEMMs.  Component SIGNS are LAPACK's choice in faiss; here each row is oriented so that its
                        largest-magnitude entry is positive.  k-means is invariant to that.
* faiss.Kmeans          Clustering::train of faiss 1.7.2 with the reference's parameters (niter 50, nredo 5, seed 1,
                        max_points_per_centroid 256, min 39): random subsample of k * 256 points, k random points as initial
                        centroids per redo, Lloyd iterations (assignment + mean), empty clusters re-seeded by splitting a
                        large one (perturbation 1/1024), best objective over the redos kept.  faiss' own RNG stream is not
                        reproduced (``torch.Generator`` / ``numpy.RandomState`` with the same seed formulas), so cluster ids
                        and, on hard data, the local optimum can differ from a faiss run: parity here is "unpinned" and the
                        tests check the algorithm against a NumPy restatement fed with the same random draws.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import hip_ops as ops


# ------------------------------------------------------------------------------------------------
# StandardScaler + PCA  (my_utils.py:19-37)
# ------------------------------------------------------------------------------------------------

def fit_scaler_pca(feats: torch.Tensor, pca_dim: int):
    """feats [n, dim] on the GPU -> (scale_mul [dim], scale_shift [dim], z_mean [dim], basis [pca_dim, dim], z [n, dim]) with
    ``z = feats * scale_mul + scale_shift`` (the standardised features) such that ``(z - z_mean) @ basis.T`` is
    ``normalize_and_transform(feats, pca_dim)``."""
    n, dim = feats.shape
    mean, var = ops.col_moments(feats)
    scale = var.sqrt()
    scale = torch.where(scale < 10 * torch.finfo(torch.float64).eps, torch.ones_like(scale), scale)  # sklearn _handle_zeros_in_scale
    mul = (1.0 / scale).float()
    shift = (-mean / scale).float()
    z = ops.affine_cols_(feats.clone(), mul, shift)
    gram, _ = ops.linear_bwd_weight(z, z, need_bias=False)                    # Z^T Z  [dim, dim]
    zmean, _ = ops.col_moments(z)
    cov = gram.double() / n - torch.outer(zmean, zmean)
    evals, evecs = torch.linalg.eigh(cov.cpu())                               # ascending
    basis = evecs[:, torch.argsort(evals, descending=True)[:pca_dim]].t().contiguous()  # [pca_dim, dim]
    lead = basis.abs().argmax(dim=1)
    basis = basis * torch.sign(basis[torch.arange(basis.shape[0]), lead]).unsqueeze(1)
    return mul, shift, zmean.float(), basis.float().to(feats.device), z


def normalize_and_transform(feats: torch.Tensor, pca_dim: int) -> torch.Tensor:
    """``my_utils.normalize_and_transform``: StandardScaler then PCA to ``pca_dim`` dims; stays on the device."""
    _, _, zmean, basis, z = fit_scaler_pca(feats.contiguous().float(), pca_dim)
    return ops.linear_fwd(z, basis, -(basis @ zmean))                         # A (z - mean_z)


# ------------------------------------------------------------------------------------------------
# k-means  (faiss.Kmeans as the reference configures it)
# ------------------------------------------------------------------------------------------------

class Kmeans:
    """``faiss.Kmeans(d, k, niter=50, nredo=5, seed=1, verbose=False, gpu=False, spherical=False)`` surface: ``train(x)``,
    ``centroids`` ([k, d] numpy, as faiss exposes them), ``assign(x) -> (dist2, labels)`` in place of ``index.search(x, 1)``."""

    def __init__(self, d: int, k: int, niter: int = 50, nredo: int = 5, seed: int = 1, verbose: bool = False, gpu: bool = False,
                 spherical: bool = False, max_points_per_centroid: int = 256):
        if spherical:
            raise NotImplementedError("spherical k-means is not used by the reference")
        self.d, self.k, self.niter, self.nredo, self.seed = d, k, niter, nredo, seed
        self.max_points_per_centroid = max_points_per_centroid
        self.centroids: Optional[np.ndarray] = None
        self._centroids_dev: Optional[torch.Tensor] = None
        self.obj: list = []

    @staticmethod
    def _perm(n: int, seed: int) -> torch.Tensor:
        return torch.randperm(n, generator=torch.Generator().manual_seed(seed % (2 ** 63)))

    def _split_empty(self, cent: torch.Tensor, counts: np.ndarray, n: int) -> int:
        """faiss split_clusters: every empty cluster takes over half of a donor picked with probability proportional to its
        size; the two copies are perturbed in opposite directions by 1/1024."""
        k, d = cent.shape
        rng = np.random.RandomState(1234)
        eps = 1.0 / 1024.0
        sign = torch.tensor([1.0 + eps if j % 2 == 0 else 1.0 - eps for j in range(d)], device=cent.device)
        nsplit = 0
        for ci in range(k):
            if counts[ci] != 0:
                continue
            if n <= k or counts.max() <= 1:
                break   # nothing left to split (n == k, or every cluster holds at most one point): faiss' loop would never accept
            cj, tries = 0, 0
            while True:  # (counts[cj] - 1) / (n - k) acceptance, as faiss
                p = (counts[cj] - 1.0) / float(n - k)
                if rng.random_sample() < p:
                    break
                cj = (cj + 1) % k
                tries += 1
                if tries > 64 * k:   # vanishing acceptance (duplicates everywhere): take the largest cluster instead of spinning
                    cj = int(np.argmax(counts))
                    break
            cent[ci] = cent[cj] * sign
            cent[cj] = cent[cj] * (2.0 - sign)
            counts[ci] = counts[cj] // 2
            counts[cj] -= counts[ci]
            nsplit += 1
        return nsplit

    def train(self, x, init_indices=None) -> float:
        """x [n, d] (GPU tensor, or numpy like faiss).  ``init_indices`` [nredo, k] optionally fixes the initial centroids
        (testing aid).  Returns the best objective."""
        x = torch.as_tensor(x, dtype=torch.float32)
        if not x.is_cuda:
            x = x.cuda()
        x = x.contiguous()
        n, d = x.shape
        k = self.k
        if n < k:
            raise RuntimeError(f"Number of training points ({n}) should be at least as large as number of clusters ({k})")
        if n > k * self.max_points_per_centroid:  # subsample_training_set
            x = x[self._perm(n, self.seed)[: k * self.max_points_per_centroid].to(x.device)].contiguous()
            n = x.shape[0]
        best_obj, best = float("inf"), None
        self.obj = []
        for redo in range(self.nredo):
            if init_indices is not None:
                idx = torch.as_tensor(init_indices[redo], dtype=torch.int64)
            else:
                idx = self._perm(n, self.seed + 1 + redo * 15486557)[:k]
            cent = x[idx.to(x.device)].clone()
            obj = float("inf")
            for _ in range(self.niter):
                labels, dist2 = ops.kmeans_assign(x, cent, return_dist=True)
                obj = float(dist2.double().sum())
                sums, counts = ops.kmeans_accumulate(x, labels, k)
                counts_h = counts.cpu().numpy().copy()
                nonempty = counts > 0
                cent = torch.where(nonempty.unsqueeze(1), (sums / counts.clamp(min=1).unsqueeze(1)).float(), cent)
                if (counts_h == 0).any():
                    self._split_empty(cent, counts_h, n)
            self.obj.append(obj)
            if obj < best_obj:
                best_obj, best = obj, cent.clone()
        self._centroids_dev = best
        self.centroids = best.cpu().numpy()
        return best_obj

    def assign(self, x):
        x = torch.as_tensor(x, dtype=torch.float32)
        if not x.is_cuda:
            x = x.cuda()
        labels, dist2 = ops.kmeans_assign(x.contiguous(), self._centroids_dev, return_dist=True)
        return dist2, labels.long()


# ------------------------------------------------------------------------------------------------
# cluster_features / proto_clustering  (clustering.py:20-117)
# ------------------------------------------------------------------------------------------------

def _kmeans_maps(points: torch.Tensor, num_clusters: int) -> torch.Tensor:
    km = Kmeans(points.shape[1], num_clusters, niter=50, nredo=5, seed=1)
    km.train(points)
    return km.assign(points)[1]


def cluster_features(features, num_clusters, feature_resolution, input_resolution, evaluation_protocol, annotations=None):
    """``clustering.cluster_features``: features [bs, fs, num_patches, dim] -> cluster maps [bs, fs, R, R] int16."""
    bs, fs, num_patches, dim = features.shape
    feats = normalize_and_transform(features.reshape(bs * fs * num_patches, dim), 50)
    dim = feats.shape[1]
    feats = feats.view(bs * fs, num_patches, dim)
    R = input_resolution
    up = ops.upsample_bilinear_tokens(feats, R).view(bs, fs, R * R, dim)      # what the reference builds frame by frame
    if evaluation_protocol == "frame-wise":
        maps = []
        for i in range(bs):
            for j in range(fs):
                k = torch.unique(annotations[i, j]).shape[0] if annotations is not None else num_clusters
                maps.append(_kmeans_maps(up[i, j], k).view(1, 1, R, R))
        out = torch.cat(maps, dim=0).view(bs, fs, R, R)
    elif evaluation_protocol == "sample-wise":
        maps = []
        for i in range(bs):
            k = torch.unique(annotations[i]).shape[0] if annotations is not None else num_clusters
            maps.append(_kmeans_maps(up[i].reshape(fs * R * R, dim), k).view(1, fs, R, R))
        out = torch.cat(maps, dim=0)
    elif evaluation_protocol == "dataset-wise":
        k = torch.unique(annotations).shape[0] if annotations is not None else num_clusters
        out = _kmeans_maps(up.reshape(bs * fs * R * R, dim), k).view(bs, fs, R, R)
    else:
        raise ValueError(f"unknown evaluation protocol {evaluation_protocol!r}")
    return out.to(torch.int16)


@torch.no_grad()
def proto_clustering(x, prototypes, input_size=14, output_size=224, num_classes=None):
    """``clustering.proto_clustering``: x [samples, num_patch, dim], prototypes [k, dim] -> assignments [samples, R, R]
    (upsampled cosine scores, arg-max; with ``num_classes`` the prototypes are first merged by k-means)."""
    sample_num, num_patches, dim = x.shape
    xn = ops.l2norm_fwd(x.reshape(sample_num * num_patches, dim).contiguous().float())
    pn = ops.l2norm_fwd(prototypes.detach().contiguous().float())
    scores = ops.linear_fwd(xn, pn).view(sample_num, num_patches, -1)
    assign = ops.upsample_argmax_f32(scores, output_size)
    if num_classes is not None:
        km = Kmeans(prototypes.shape[1], num_classes, niter=50, nredo=5, seed=1)
        km.train(prototypes.detach().float())
        proto_maps = km.assign(prototypes.detach().float())[1]
        assign = proto_maps[assign.reshape(-1)].view(sample_num, output_size, output_size)
    return assign
