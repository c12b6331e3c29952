"""``sinkhorn`` and ``cosine_scheduler`` with the reference's signatures (``my_utils.py:246-283``)."""
from __future__ import annotations

import numpy as np
import torch

from . import engine
from . import hip_ops as ops


@torch.no_grad()
def sinkhorn(Q: torch.Tensor, nmb_iters: int, world_size=1) -> torch.Tensor:
    """``my_utils.py:246-274``: Q is ``exp(scores / eps).T`` of shape [K, B_local]; returns the [B_local, K] assignment.
    The positive matrix goes to the kernel as it is (``tt_sinkhorn_from_q``).  For ``world_size > 1`` the local columns are
    all-gathered once and the global problem is solved on every rank (identical to the reference's 1 + 1 + nmb_iters
    all-reduces)."""
    Q = Q.detach().float().contiguous()
    dist = engine.exchange_group() if world_size > 1 else None
    if dist is None:
        return ops.sinkhorn_from_q(Q, int(nmb_iters))
    W, (K, B) = dist.get_world_size(), Q.shape
    cols = Q.t().contiguous()                                    # [B_local, K]: rank-major rows after the gather
    gathered = torch.empty((W * B, K), dtype=Q.dtype, device=Q.device)
    try:
        dist.all_gather_into_tensor(gathered, cols)
    except RuntimeError:  # backends without the flat all-gather
        dist.all_gather(list(gathered.chunk(W, dim=0)), cols)
    return ops.sinkhorn_from_q(gathered, int(nmb_iters), row0=dist.get_rank() * B, rows_out=B, transposed=True)


def cosine_scheduler(base_value: float, final_value: float, epochs: int, niter_per_ep: int):
    """``my_utils.py:278-283``."""
    iters = np.arange(epochs * niter_per_ep)
    schedule = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    assert len(schedule) == epochs * niter_per_ep
    return schedule
