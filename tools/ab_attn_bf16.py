#!/usr/bin/env python3
"""A/B of tt_attention_fwd_bf16 between library builds in one process. usage: ab_attn_bf16.py libA.so libB.so ... [F N H]"""
import ctypes as C, os, statistics, sys, torch
def load(p):
    lib = C.CDLL(os.path.abspath(p)); lib.tt_attention_fwd_bf16.restype = C.c_int
    lib.tt_attention_fwd_bf16.argtypes = [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_float, C.c_void_p]; return lib
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
dims = [int(a) for a in sys.argv[1:] if not a.endswith(".so")]
F, N, H = dims if len(dims) == 3 else (128, 197, 12)
libs = [(p, load(p)) for p in paths]
qkv = torch.randn(F, N, 3 * H * 64, device="cuda").to(torch.bfloat16); out = torch.empty(F, N, H * 64, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
res = {p: [] for p, _ in libs}
for rd in range(12):
    for p, lib in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): assert lib.tt_attention_fwd_bf16(qkv.data_ptr(), out.data_ptr(), F, N, H, 64, 0.125, st) == 0
        e1.record(); torch.cuda.synchronize()
        if rd >= 2: res[p].append(e0.elapsed_time(e1) * 1e2)
outs = {}
for p, lib in libs:
    out.zero_(); lib.tt_attention_fwd_bf16(qkv.data_ptr(), out.data_ptr(), F, N, H, 64, 0.125, st); torch.cuda.synchronize(); outs[p] = out.float().clone()
p0 = libs[0][0]
for p, _ in libs[1:]: print(f"max |out[{os.path.basename(p)}] - out[{os.path.basename(p0)}]| = {(outs[p] - outs[p0]).abs().max().item():.3e}  (max |out| {outs[p0].abs().max().item():.3f})")
for p, v in res.items(): print(f"{os.path.basename(p):28s} median {statistics.median(v):7.1f} us  min {min(v):7.1f} us")
