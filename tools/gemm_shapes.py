#!/usr/bin/env python3
"""Times tt_linear_fwd over shapes given as M,N,K[,act] arguments (development aid)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops
if os.environ.get('TT_PRECISION'): ops.set_gemm_precision(os.environ['TT_PRECISION'])
for arg in sys.argv[1:]:
    p = [int(v) for v in arg.split(",")]
    M, N, K = p[:3]; act = p[3] if len(p) > 3 else 0
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.zeros(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    ts = []
    for rd in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.linear_fwd(x, w, b if act >= 0 else None, act=max(act, 0), out=y)
        e1.record(); torch.cuda.synchronize()
        if rd >= 2: ts.append(e0.elapsed_time(e1) * 1e-3 / 3)
    t = statistics.median(ts)
    print(f"M={M} N={N} K={K} act={act}: {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:6.1f} TF")
