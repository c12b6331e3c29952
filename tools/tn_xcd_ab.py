#!/usr/bin/env python3
"""gemm_pairs_tn: the workgroups of a split on one XCD (knob TT_TN_XCD) - interleaved A/B with the split target swept, outputs against fp64."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)
for M, N, K in [(6304, 1152, 384), (6304, 384, 384), (6304, 1536, 384), (6304, 384, 1536), (6272, 1024, 1024), (12544, 1536, 384)]:
    dy = torch.randn(M, N, device="cuda") * 0.02; x = torch.randn(M, K, device="cuda")
    xp = ops.split_pairs(x); dyr = ops.split_pairs(dy); ref = dy.double().t() @ x.double(); out = []; worst = 0.0
    for xcd in (0, 1):
        for w in (0, 256, 512):
            ops.set_tuning_knob("TT_TN_XCD", xcd); ops.set_tuning_knob("TT_TN_WGS", w)
            d = ops.linear_bwd_weight_pairs_tn(dyr, xp)
            worst = max(worst, ((d.double() - ref).norm() / ref.norm()).item())
            out.append(f"xcd{xcd} wgs{w}: {t(lambda: ops.linear_bwd_weight_pairs_tn(dyr, xp)):6.1f}")
    print(M, N, K, " | ".join(out), f"  worst rel-L2 {worst:.1e}", flush=True)
ops.set_tuning_knob("TT_TN_XCD", 1); ops.set_tuning_knob("TT_TN_WGS", 0)
