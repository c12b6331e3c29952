#!/usr/bin/env python3
"""Times tt_linear_fwd_planes on the block shapes of C2 (ViT-S/16, planes 3) and C4 (ViT-B/16, planes 1).
TT_PLANES_VARIANT=<n> selects a tuning variant of the dispatcher (gemm_planes.hip); one process per variant."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops

def bench(P, M, N, K, act=0, out_planes=0, reps=20):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    xp, wp = ops.split_planes(x, P), ops.split_planes(w, P)
    for _ in range(3): ops.linear_fwd_planes(xp, wp, b, act=act, out_f32=out_planes == 0, out_planes=out_planes)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ops.linear_fwd_planes(xp, wp, b, act=act, out_f32=out_planes == 0, out_planes=out_planes)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / reps)
    t = statistics.median(ts)
    return t * 1e6, 2.0 * M * N * K / t / 1e12

v = os.environ.get("TT_PLANES_VARIANT", "0")
M = 25216
for P, shapes in ((3, [(1152, 384, "qkv"), (384, 384, "proj"), (1536, 384, "fc1"), (384, 1536, "fc2")]),
                  (1, [(2304, 768, "qkv"), (768, 768, "proj"), (3072, 768, "fc1"), (768, 3072, "fc2")])):
    for N, K, name in shapes:
        us, tf = bench(P, M, N, K, act=1 if name == "fc1" else 0, out_planes=P if name in ("fc1",) else 0)
        print(f"variant {v}  P={P} {name:5s} M={M} N={N:5d} K={K:5d}  {us:8.1f} us  {tf:7.1f} TFLOP/s-equivalent  (raw bf16 MFMA rate x{P*(P+1)//2}: {tf*P*(P+1)//2:7.0f})", flush=True)
