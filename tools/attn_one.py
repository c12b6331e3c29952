#!/usr/bin/env python3
"""Runs tt_attention_fwd a few times on one shape (for rocprofv3 --pmc passes): attn_one.py F N H"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops
F_, N, H = [int(v) for v in sys.argv[1:4]]
qkv = torch.randn(F_, N, 3 * H * 64, device="cuda")
for _ in range(6): ops.attention_fwd(qkv, H)
torch.cuda.synchronize()
