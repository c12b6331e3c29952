#!/usr/bin/env python3
"""Sinkhorn at sizes where E no longer fits the caches (SURVEY.md 8(d) stress variant): HBM GB/s of tt_sinkhorn.
Algorithmic bytes per call = 4 K B (2 iters + 2): scores read once, E written once and read twice per iteration pair, q written."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops

for K, B in ((200, 6272), (200, 50176), (200, 401408), (400, 1 << 20)):
    sc = (torch.rand(B, K, device="cuda") - 0.5) * 0.4
    for _ in range(2):
        ops.sinkhorn(sc, 10)
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); q = ops.sinkhorn(sc, 10); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = statistics.median(ts)
    byts = 4.0 * K * B * 22
    rows = q.sum(1)
    print(f"K={K} B={B}: E = {4.0 * K * B / 1e6:8.1f} MB, {t * 1e6:9.1f} us per 10-iteration solve = {10 / t:9.0f} it/s, "
          f"{byts / t / 1e9:7.0f} GB/s algorithmic ({byts / t / 8e12:.2f} of 8 TB/s); row sums in [{rows.min().item():.6f}, {rows.max().item():.6f}]")
    del sc, q
