"""``sinkhorn`` and ``cosine_scheduler`` with the reference's signatures (``my_utils.py:246-283``)."""
from __future__ import annotations

import numpy as np
import torch

from . import engine
from . import hip_ops as ops


@torch.no_grad()
def sinkhorn(Q: torch.Tensor, nmb_iters: int, world_size=1) -> torch.Tensor:
    """``my_utils.py:246-274``: Q is ``exp(scores / eps).T`` of shape [K, B_local]; returns the [B_local, K] assignment.

    The kernel works on scores, so the positive matrix is mapped back with ``log`` (eps = 1): exp(log Q) = Q.  For
    ``world_size > 1`` the local columns are all-gathered once and the global problem is solved on every rank
    (identical to the reference's 1 + 1 + nmb_iters all-reduces)."""
    scores = torch.log(Q.detach().float()).t().contiguous()
    if world_size > 1:
        return engine.global_sinkhorn(scores, scores.shape[0], 1.0, int(nmb_iters))
    return ops.sinkhorn(scores, int(nmb_iters), 1.0)


def cosine_scheduler(base_value: float, final_value: float, epochs: int, niter_per_ep: int):
    """``my_utils.py:278-283``."""
    iters = np.arange(epochs * niter_per_ep)
    schedule = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    assert len(schedule) == epochs * niter_per_ep
    return schedule
