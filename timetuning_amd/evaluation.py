"""Evaluator with the reference's surface (``evaluation.py:250-310,312-486``): features without head -> clustering ->
matched mIoU, for the three protocols (frame-wise / sample-wise / dataset-wise).  Dataset readers, video / GIF logging and
the wandb plumbing are out of scope; the loader is any iterable of ``(data, annotations[, label])`` batches."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .clustering import cluster_features, proto_clustering
from .metrics import PredsmIoU
from .models import FeatureExtractor


def evaluate_localizations(PredsEval, gts, preds, evaluation_protocol, logging_directory=None, many_to_one=False, precision_based=False):
    """gts / preds [bs, fs, R, R] -> mean score (``evaluation.py:250-310``)."""
    if logging_directory is not None:
        raise NotImplementedError("video logging of the matched maps is not part of this build")
    scores = []
    if evaluation_protocol == "frame-wise":
        for i, datum in enumerate(preds):
            for j, frame in enumerate(datum):
                PredsEval.update(gts[i, j].flatten(), frame.flatten())
                scores.append(PredsEval.compute(True, many_to_one, precision_based=precision_based)[0])
                PredsEval.reset()
    elif evaluation_protocol == "sample-wise":
        for i, datum in enumerate(preds):
            for j, frame in enumerate(datum):
                PredsEval.update(gts[i, j].flatten(), frame.flatten())
            scores.append(PredsEval.compute(True, many_to_one, precision_based=precision_based)[0])
            PredsEval.reset()
    elif evaluation_protocol == "dataset-wise":
        for i, datum in enumerate(preds):
            for j, frame in enumerate(datum):
                valid = gts[i, j] != 255   # Pascal VOC's ignore label (:304-305)
                PredsEval.update(gts[i, j][valid].flatten(), frame[valid].flatten())
        scores.append(PredsEval.compute(True, many_to_one, precision_based=precision_based)[0])
        PredsEval.reset()
    else:
        raise ValueError(f"unknown evaluation protocol {evaluation_protocol!r}")
    return sum(scores) / len(scores)


class Evaluator:
    """``Evaluator(model, data_loader, num_prototypes, device, logger, clustering_algorithm)`` (``evaluation.py:312-371``) reduced
    to what ``evaluate`` needs."""

    def __init__(self, model, data_loader, num_prototypes=21, device="cuda", logger=None, clustering_algorithm="k-means", uvos_flag=False,
                 involve_bg=False):
        self.model, self.data_loader, self.device = model, data_loader, device
        self.clustering_algorithm = clustering_algorithm
        self.uvos_flag = uvos_flag
        self.PredsEval = PredsmIoU(num_prototypes, num_prototypes, involve_bg=involve_bg)

    def _features(self, data):
        fe = self.model if isinstance(self.model, FeatureExtractor) else self.model.feature_extractor
        bs, fs, c, h, w = data.shape
        feats, _ = fe(data.view(bs * fs, c, h, w).to(self.device), use_head=False)
        return feats.view(bs, fs, feats.shape[1], feats.shape[2]), fe.spatial_resolution

    def _cluster(self, features, spatial_resolution, eval_resolution, protocol, num_clusters, annotations):
        if self.clustering_algorithm == "k-means":
            return cluster_features(features, num_clusters, spatial_resolution, eval_resolution, protocol, annotations)
        if self.clustering_algorithm == "prototypes":
            bs, fs, n, dim = features.shape
            maps = proto_clustering(features.view(bs * fs, n, dim), self.model.prototypes, spatial_resolution, output_size=eval_resolution,
                                    num_classes=num_clusters)
            return maps.view(bs, fs, eval_resolution, eval_resolution)
        raise ValueError(f"unknown clustering algorithm {self.clustering_algorithm!r}")

    @torch.no_grad()
    def evaluate(self, many_to_one=False, evaluation_protocol="frame-wise", eval_resolution=None, num_clusters=10, use_mask=False,
                 use_annotations=False, precision_based=False):
        """``evaluation.py:373-480``.  Batches are ``(data [bs,(1,)fs,3,H,W], annotations [bs,(1,)fs,H,W] integer labels[, label])``."""
        if use_mask:
            raise NotImplementedError("evaluation on attention-masked features is not part of this build")
        self.model.eval()
        if evaluation_protocol == "dataset-wise":
            feats, anns = [], []
            for batch in self.data_loader:
                data, annotations = batch[0], batch[1]
                if data.dim() == 6:
                    data, annotations = data.squeeze(1), annotations.squeeze(1)
                f, g = self._features(data)
                feats.append(f)
                anns.append(annotations.long())
            features, annotations = torch.cat(feats, dim=0), torch.cat(anns, dim=0)
            annotations = F.interpolate(annotations.double(), size=(eval_resolution, eval_resolution), mode="nearest").long().to(self.device)
            maps = self._cluster(features, g, eval_resolution, evaluation_protocol, num_clusters, annotations if use_annotations else None)
            return evaluate_localizations(self.PredsEval, annotations, maps, evaluation_protocol, None, many_to_one, precision_based)
        scores = []
        for batch in self.data_loader:
            data, annotations = batch[0].squeeze(1), batch[1].squeeze(1).long()
            features, g = self._features(data)
            if self.uvos_flag:
                annotations = (annotations > 0).long()
            annotations = F.interpolate(annotations.double(), size=(eval_resolution, eval_resolution), mode="nearest").long().to(self.device)
            maps = self._cluster(features, g, eval_resolution, evaluation_protocol, num_clusters, annotations if use_annotations else None)
            scores.append(evaluate_localizations(self.PredsEval, annotations, maps, evaluation_protocol, None, many_to_one, precision_based))
        return sum(scores) / len(scores)
