#!/usr/bin/env python3
"""Runs tt_linear_fwd_planes a few times on one shape (for rocprofv3 --pmc passes): planes_one.py P M N K [act] [out_planes] [residual]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops
P, M, N, K = [int(v) for v in sys.argv[1:5]]
act = int(sys.argv[5]) if len(sys.argv) > 5 else 0
po = int(sys.argv[6]) if len(sys.argv) > 6 else 0
res = int(sys.argv[7]) if len(sys.argv) > 7 else 0
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
xp, wp = ops.split_planes(x, P), ops.split_planes(w, P)
r = torch.randn(M, N, device="cuda") if res else None
for _ in range(6): ops.linear_fwd_planes(xp, wp, b, residual=r, act=act, out_f32=po == 0, out_planes=po, out=r)
torch.cuda.synchronize()
