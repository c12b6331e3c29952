import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from timetuning_amd import hip_ops as ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)
for M, N, K in [(6304, 1152, 384), (6304, 384, 384), (6304, 1536, 384), (6304, 384, 1536), (6304, 768, 3072)]:
    dy = torch.randn(M, N, device="cuda") * 0.02; x = torch.randn(M, K, device="cuda")
    xp = ops.split_pairs(x); dyr = ops.split_pairs(dy)
    out = []
    for w in (64, 128, 192, 256, 384, 512, 768):
        ops.set_tuning_knob("TT_TN_WGS", w)
        out.append(f"{w}: {t(lambda: ops.linear_bwd_weight_pairs_tn(dyr, xp)):6.1f}")
    print(M, N, K, " | ".join(out), flush=True)
