#!/usr/bin/env python3
"""Times tt_linear_fwd (fp32) on the ViT-S/16 block shapes of C2 and checks it against fp64.  Env switches of the library
(TT_GEMM_DMA=<ring depth>, TT_FORCE_TILE) select variants; one process per variant."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops

tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("TT_"))
M = int(os.environ.get("BENCH_M", "25216"))
tot = 0.0
for N, K, name, act, res in ((1152, 384, "qkv", 0, False), (384, 384, "proj", 0, True), (1536, 384, "fc1", 1, False), (384, 1536, "fc2", 0, True)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    y = ops.linear_fwd(x, w, b, residual=r, act=act)
    ref = torch.nn.functional.linear(x[:512].double(), w.double(), b.double())
    if act: ref = torch.nn.functional.gelu(ref)
    if res: ref = ref + r[:512].double()
    err = ((y[:512].double() - ref).abs().max() / ref.abs().max()).item()
    for _ in range(3): ops.linear_fwd(x, w, b, residual=r, act=act)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.linear_fwd(x, w, b, residual=r, act=act)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / 20)
    t = statistics.median(ts); tot += t
    print(f"[{tag}] {name:5s} M={M} N={N:5d} K={K:5d}  {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:6.1f} TFLOP/s  rel err {err:.1e}", flush=True)
print(f"[{tag}] block total {tot*1e6:8.1f} us", flush=True)
