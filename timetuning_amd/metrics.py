"""``PredsmIoU`` with the reference's surface (``metrics.py:209-505``): matched mean IoU between predicted cluster maps and
ground-truth label maps, Hungarian or many-to-one matching.

The reference concatenates every prediction / ground-truth pixel on the host and re-scans the two arrays once per
(ground-truth class, predicted class) pair (``compute_score_matrix``, joblib-parallel).  Every quantity it derives - the
score matrix, the matching, tp / fp / fn after remapping - is a function of the confusion matrix alone, so here one GPU
pass (``tt_confusion_counts``) builds that matrix and the rest is exact integer bookkeeping on a few hundred numbers.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import hip_ops as ops


class PredsmIoU(torch.nn.Module):
    def __init__(self, num_pred_classes: int, num_gt_classes: int, involve_bg: bool = False):
        super().__init__()
        self.num_pred_classes = num_pred_classes
        self.num_gt_classes = num_gt_classes
        self.gt: List[torch.Tensor] = []
        self.pred: List[torch.Tensor] = []
        self.involve_bg = involve_bg

    def update(self, gt: torch.Tensor, pred: torch.Tensor) -> None:
        self.gt.append(gt.reshape(-1))
        self.pred.append(pred.reshape(-1))

    def reset(self) -> None:
        self.gt, self.pred = [], []

    # -- confusion matrix over the VALUES that occur (the reference re-derives the class sets from the data, :262-263)
    def _confusion(self):
        pred = torch.cat(self.pred).long().cuda().contiguous()
        gt = torch.cat(self.gt).long().cuda().contiguous()
        C = int(max(pred.max().item(), gt.max().item())) + 1
        counts = ops.confusion_counts(pred, gt, C).cpu().numpy().astype(np.int64)   # [gt value, pred value]
        gt_unique = np.nonzero(counts.sum(1))[0]
        pred_unique = np.nonzero(counts.sum(0))[0]
        return pred, counts[np.ix_(gt_unique, pred_unique)], gt_unique, pred_unique

    def compute(self, is_global_zero: bool, many_to_one: bool = False, precision_based: bool = False, linear_probe: bool = False):
        if not is_global_zero:
            return None
        pred, conf, gt_unique, pred_unique = self._confusion()
        self.num_pred_classes, self.num_gt_classes = len(pred_unique), len(gt_unique)
        return self.compute_miou_from_confusion(conf, gt_unique, pred_unique, pred, many_to_one, precision_based, linear_probe)

    @staticmethod
    def score_matrix(conf: np.ndarray, precision_based: bool = False) -> np.ndarray:
        """[num_gt, num_pred]: IoU (or precision) if ground-truth class i were matched to predicted class j (:435-474)."""
        tp = conf.astype(np.float64)
        fp = conf.sum(0, keepdims=True) - tp
        if precision_based:
            return tp / np.maximum(tp + fp, 1e-8)
        fn = conf.sum(1, keepdims=True) - tp
        return tp / np.maximum(tp + fp + fn, 1e-8)

    def compute_miou_from_confusion(self, conf, gt_unique, pred_unique, pred=None, many_to_one=False, precision_based=False,
                                    linear_probe=False):
        """``compute_miou`` (:357-432) on the confusion matrix ``conf[gt index, pred index]``."""
        num_gt, num_pred = conf.shape
        mapping: Dict[int, int] = {}          # predicted VALUE -> ground-truth VALUE; unmapped predictions count as 0
        if linear_probe:
            mapping = {int(p): int(p) for p in pred_unique}
            matched_bg_clusters = {}
        elif many_to_one:
            match = self._original_match(conf, precision_based)
            for target_i, matched_preds in match.items():
                for pred_i in matched_preds:
                    mapping[int(pred_unique[pred_i])] = int(gt_unique[target_i])
            matched_bg_clusters = len(match[0]) / num_pred
        else:
            rows, cols = linear_sum_assignment(1 - self.score_matrix(conf))
            for target_i, pred_i in zip(rows, cols):
                mapping[int(pred_unique[pred_i])] = int(gt_unique[target_i])
            matched_bg_clusters = 1 / num_gt
        # confusion after remapping: remapped[gt index][target VALUE]
        tp, fp, fn, jac = {}, {}, {}, {}
        col_target = np.array([mapping.get(int(p), 0) for p in pred_unique])
        for gi, g in enumerate(gt_unique):
            sel = col_target == g
            tp_g = int(conf[gi, sel].sum())
            fp_g = int(conf[:, sel].sum()) - tp_g
            fn_g = int(conf[gi].sum()) - tp_g
            tp[int(g)], fp[int(g)], fn[int(g)] = tp_g, fp_g, fn_g
            jac[int(g)] = float(tp_g) / max(float(tp_g + fp_g + fn_g), 1e-8)
        if not self.involve_bg:
            jac.pop(0, None)
            if len(jac) == 0:  # the found cluster is solely background (:426-427)
                jac[0] = 0
        reordered = None
        if pred is not None:
            lut = torch.zeros(int(pred_unique.max()) + 1, dtype=torch.int64)
            for p, g in mapping.items():
                lut[p] = g
            reordered = lut.to(pred.device)[pred]
        return float(np.mean(np.array(list(jac.values())))), tp, fp, fn, reordered, matched_bg_clusters

    def _original_match(self, conf, precision_based=False) -> Dict[int, list]:
        """Greedy many-to-one: every predicted class goes to the ground-truth class with the best score (:489-505)."""
        score_mat = self.score_matrix(conf, precision_based)
        gt_to_matches = defaultdict(list)
        for pred_c in range(conf.shape[1]):
            gt_to_matches[int(np.argmax(score_mat[:, pred_c]))].append(pred_c)   # first maximum, as the `>` scan
        return gt_to_matches
