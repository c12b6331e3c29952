#!/usr/bin/env python3
"""gemm_planes8 (P = 1) K-split of the left-over tiles (knob TT_Q8_KSPLIT, shared with gemm_pairs8): correctness against fp64 and the unsplit
kernel, run-to-run bit equality, interleaved A/B.   python tools/p8_ksplit.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops
knob = ops.set_tuning_knob
bad = 0
for M, N, K, res, name in [(25216, 768, 3072, 1, "B fc2"), (25216, 768, 768, 1, "B proj"), (25216, 768, 3072, 0, "B fc2, no residual"), (25000, 768, 1536, 1, "ragged"),
                            (25216, 256, 1536, 1, "S fc2 (N 256)"), (25216, 2304, 768, 0, "B qkv")]:
    torch.manual_seed(3)
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda") * 0.1
    r0 = torch.randn(M, N, device="cuda") if res else None
    xp, wp = ops.split_planes(x, 1), ops.split_planes(w, 1)
    def go(mode):
        knob("TT_Q8_KSPLIT", mode)
        r = r0.clone() if res else None
        return ops.linear_fwd_planes(xp, wp, b, residual=r, act=0, out_f32=True, out_planes=0, out=r)["y"]
    idx = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,))]).cuda()
    ref = xp.double().sum(0)[idx] @ wp.double().sum(0).t() + b.double()
    if res: ref = ref + r0.double()[idx]
    o0, o1 = go(0), go(1)
    e0, e1 = [((o.double()[idx] - ref).norm() / ref.norm()).item() for o in (o0, o1)]
    full = ((o1 - o0).abs().max() / o0.abs().max()).item()
    same = all(torch.equal(go(1), o1) for _ in range(6))
    ts = {0: [], 1: []}
    for rd in range(8):
        for m in ts:
            knob("TT_Q8_KSPLIT", m)
            r = r0.clone() if res else None
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(10): ops.linear_fwd_planes(xp, wp, b, residual=r, act=0, out_f32=True, out_planes=0, out=r)
            a1.record(); torch.cuda.synchronize()
            if rd >= 2: ts[m].append(a0.elapsed_time(a1) * 1e2)
    ok = same and full < 2e-5 and e0 < 1e-5 and e1 < 1e-5
    bad += not ok
    print(f"{name:20s} M={M} N={N} K={K}: rel-L2 vs fp64 (of the bf16 operands) off {e0:.2e} split {e1:.2e} | split-off max {full:.1e} | repeat {'ok' if same else 'DIFFERS'} | "
          f"off {statistics.median(ts[0]):7.1f} us | on {statistics.median(ts[1]):7.1f} us" + ("" if ok else "   FAIL"), flush=True)
knob("TT_Q8_KSPLIT", 1)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
