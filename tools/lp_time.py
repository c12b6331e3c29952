#!/usr/bin/env python3
"""Times tt_label_propagate at a training shape in a precision mode (the library reads TT_LP_BK once per process: run it twice to A/B).
usage: lp_time.py [precision=bf16] [bs=16] [fs=8] [D=768] [K=400]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from timetuning_amd import hip_ops as ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
bs, fs, D, K = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (16, 8, 768, 400)))
n = 196
torch.manual_seed(0)
xn = F.normalize(torch.randn(fs, bs, n, D, device="cuda"), dim=-1)
seg0 = torch.softmax(torch.randn(bs, n, K, device="cuda") * 2, -1)
ops.set_gemm_precision(prec)
ts = []
for rd in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): lab = ops.label_propagate(xn, seg0, 7, 6, 5, 0.1)
    e1.record(); torch.cuda.synchronize()
    if rd >= 2: ts.append(e0.elapsed_time(e1) * 200)
print(f"label_propagate {prec} bs={bs} fs={fs} D={D} K={K} TT_LP_BK={os.environ.get('TT_LP_BK', '(default)')}: median {statistics.median(ts):.1f} us  labels checksum {int(lab.sum())}")
