#!/usr/bin/env python3
"""gemm_pairs8 K-split of the left-over tiles (knob TT_Q8_KSPLIT: 0 off, 1 behind whole rounds, 2 also grids under one round):
correctness against fp64 and against the unsplit kernel, run-to-run bit equality (the slice order of the sum is fixed), and an
interleaved A/B in one process.   python tools/q8_ksplit.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops

knob = ops.set_tuning_knob
cases = [  # M, N, K, act, pairs out, residual, name
    (25216, 1152, 384, 0, 1, 0, "qkv"), (25216, 384, 384, 0, 0, 1, "proj"), (25216, 1536, 384, 1, 1, 0, "fc1"), (25216, 384, 1536, 0, 0, 1, "fc2"),
    (25216, 768, 768, 0, 0, 1, "B proj"), (25216, 768, 3072, 0, 0, 1, "B fc2"), (25216, 2304, 768, 0, 1, 0, "B qkv"),
    (6304, 384, 384, 0, 0, 1, "kept proj"), (6304, 384, 1536, 0, 0, 1, "kept fc2"), (6304, 384, 1152, 0, 0, 0, "dgrad qkv"), (6304, 1152, 384, 0, 1, 0, "kept qkv"),
    (18912, 384, 1536, 0, 0, 1, "rest fc2"), (18912, 384, 384, 0, 0, 1, "rest proj"), (6500, 384, 384, 0, 0, 1, "ragged"), (50177, 384, 768, 0, 0, 0, "ragged2"),
    (100480, 384, 1536, 0, 0, 1, "C5 fc2")]
bad = 0
for M, N, K, act, po, res, name in cases:
    torch.manual_seed(1)
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda") * 0.1
    r0 = torch.randn(M, N, device="cuda") if res else None
    xp, wp = ops.split_pairs(x), ops.split_pairs(w)
    def go(mode):
        knob("TT_Q8_KSPLIT", mode)
        r = r0.clone() if res else None
        o = ops.linear_fwd_pairs(xp, wp, b, residual=r, act=act, out_f32=not po, out_pairs=bool(po), out=r)
        return o["y"] if not po else ops.join_pairs(o["pairs"])
    idx = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,))]).cuda()
    ref = x.double()[idx] @ w.double().t() + b.double()
    if act: ref = torch.nn.functional.gelu(ref)
    if res: ref = ref + r0.double()[idx]
    MODES = (0, 1, 31, 41, 2)
    outs = {m: go(m) for m in (0, 1, 2)}
    err = {m: ((outs[m].double()[idx] - ref).norm() / ref.norm()).item() for m in outs}
    full = ((outs[2] - outs[0]).abs().max() / outs[0].abs().max()).item()
    same = all(torch.equal(go(2), outs[2]) for _ in range(6))
    routes = []
    for m in (0, 2):
        knob("TT_Q8_KSPLIT", m)
        routes.append(ops._lib.load().tt_linear_fwd_pairs_route(M, N, K, act, 1, int(bool(res)), int(not po), int(bool(po)), 0))
    ts = {m: [] for m in MODES}
    for rd in range(8):
        for m in ts:
            knob("TT_Q8_KSPLIT", m)
            r = r0.clone() if res else None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.linear_fwd_pairs(xp, wp, b, residual=r, act=act, out_f32=not po, out_pairs=bool(po), out=r)
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: ts[m].append(e0.elapsed_time(e1) * 1e2)
    ok = same and full < 2e-6 and all(e < 6e-7 for e in err.values())
    bad += not ok
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d} route {routes}: rel-L2 vs fp64 off {err[0]:.2e} split {err[2]:.2e} | split-off max {full:.1e} | repeat {'ok' if same else 'DIFFERS'} | "
          + " | ".join(f"mode {m} {statistics.median(v):7.1f} us" for m, v in ts.items()) + ("" if ok else "   FAIL"), flush=True)
knob("TT_Q8_KSPLIT", 1)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
