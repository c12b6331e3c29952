#!/usr/bin/env python3
"""Attention backward: the fp32-MFMA kernels against tt_attention_bwd_pairs (S and dP on fp16 pairs), us per call and error against fp64."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timetuning_amd import hip_ops as ops
for F, N, H in [(32, 197, 6), (16, 197, 12), (16, 785, 6), (128, 197, 6)]:
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(F, N, 3 * H * 64, generator=g) * 1.2).cuda()
    do = (torch.randn(F, N, H * 64, generator=g) * 1e-4).cuda()
    out, lse, _ = ops.attention_fwd(qkv, H, save_lse=True)
    res = {}
    for name, kw in (("f32", {}), ("pairs", dict(pair_products=True))):
        ts = []
        for rd in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): d = ops.attention_bwd(qkv, out, do, lse, H, **kw)
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: ts.append(e0.elapsed_time(e1) * 200)
        res[name] = (statistics.median(ts), d)
    if F * N <= 8000:
        qd = qkv[:2].double().cpu().requires_grad_(True)
        t = qd.reshape(2, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        o = (torch.softmax(t[0] @ t[1].transpose(-1, -2) * 0.125, -1) @ t[2]).transpose(1, 2).reshape(2, N, H * 64)
        (o * do[:2].double().cpu()).sum().backward()
        err = {k: ((v[1][:2].double().cpu() - qd.grad).norm() / qd.grad.norm()).item() for k, v in res.items()}
    else:
        err = {k: float("nan") for k in res}
    print(f"F={F} N={N} H={H}: " + " | ".join(f"{k} {v[0]:7.1f} us (rel L2 {err[k]:.2e})" for k, v in res.items()), flush=True)
ops.check_pair_range()
