#!/usr/bin/env python3
"""Every matrix-product launch of one C2 training step with its shape and duration (HIP events around each launch, as bench.py's
instrumented step): which shapes the small-tile kernels carry.   python tools/step_gemms.py [--precision f16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from timetuning_amd import hip_ops as ops, synth
from timetuning_amd.my_utils import cosine_scheduler
from timetuning_amd.time_tuning import SwavOptimizer

prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "f16x3"
ops.set_gemm_precision(prec)
dev = torch.device("cuda", 0)
model = bench.build_model("dino-s16", 200, dev)
opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 100), 100, 1)
x = torch.from_numpy(synth.make_clips(32, 4, 224, seed=1)).to(dev)
for _ in range(3): bench.train_step(model, opt, x, False)
shapes = []
real = ops._prof_end
def wrapped(e0, name, M, N, K, batch=1):
    n0 = len(ops.PROFILE) if ops.PROFILE is not None else 0
    real(e0, name, M, N, K, batch)
    if ops.PROFILE is not None and len(ops.PROFILE) > n0: shapes.append((M, N, K, batch))
ops._prof_end = wrapped
rec = []
ops.PROFILE = rec
bench.train_step(model, opt, x, False); torch.cuda.synchronize()
ops.PROFILE = None
rows = {}
for (name, tile, flops, e0, e1), (M, N, K, b) in zip(rec, shapes):
    d = rows.setdefault((name, M, N, K, b), [0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in rows.values())
print(f"{len(rec)} launches, {tot / 1e3:.2f} ms in matrix products")
for (name, M, N, K, b), (c, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d} x{b}: {c:3d} launches, {us / c:7.1f} us each, {us:8.1f} us  ({2.0 * M * N * K * b * c / us * 1e-6:6.1f} TFLOP/s-equivalent)")
