#!/usr/bin/env python3
"""One kernel of the "f16x3" mode, a few launches, for the rocprofv3 --pmc passes (tools/pmc_pairs.sh).
  pairs_one.py gemm M N K epi      epi: f32 | res | pairs | gelu        (gemm_pairs8_kernel where the shape allows)
  pairs_one.py attn F N H          attention_fwd_pairs_kernel (N <= 256) / attention_fwd_pairs_flash_kernel
  pairs_one.py tn M N K            gemm_pairs_tn_kernel: dW [N, K] from row pairs dy [M, N], x [M, K]"""
import sys
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from timetuning_amd import hip_ops as ops  # noqa: E402

kind = sys.argv[1]
torch.manual_seed(0)
if kind == "gemm":
    M, N, K, epi = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    xp, wp = ops.split_pairs(torch.randn(M, K, device="cuda")), ops.split_pairs(torch.randn(N, K, device="cuda") * 0.05)
    b, y = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    fn = {"f32": lambda: ops.linear_fwd_pairs(xp, wp, b, out=y), "res": lambda: ops.linear_fwd_pairs(xp, wp, b, residual=y, out=y),
          "pairs": lambda: ops.linear_fwd_pairs(xp, wp, b, out_f32=False, out_pairs=True),
          "gelu": lambda: ops.linear_fwd_pairs(xp, wp, b, act=1, out_f32=False, out_pairs=True)}[epi]
elif kind == "tn":
    M, N, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dyp, xp = ops.split_pairs(torch.randn(M, N, device="cuda")), ops.split_pairs(torch.randn(M, K, device="cuda"))
    fn = lambda: ops.linear_bwd_weight_pairs_tn(dyp, xp)
else:
    F, N, H = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    qkvp = ops.split_pairs(torch.randn(F * N, 3 * 64 * H, device="cuda")).view(F, N, 6 * 64 * H)
    fn = lambda: ops.attention_fwd_pairs(qkvp, H)
for _ in range(12):
    fn()
torch.cuda.synchronize()
