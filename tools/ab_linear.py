#!/usr/bin/env python3
"""A/B of tt_linear_fwd (fp32) between library builds on the ViT-S/16 block shapes: usage ab_linear.py libA.so libB.so ..."""
import ctypes as C, os, statistics, sys, torch
vp, i32 = C.c_void_p, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p)); lib.tt_linear_fwd.restype = C.c_int
    lib.tt_linear_fwd.argtypes = [vp] * 6 + [i32] * 5 + [vp]; return lib
libs = [(p, load(p)) for p in sys.argv[1:]]
M = 25216; st = torch.cuda.current_stream().cuda_stream
for N, K, name, act, res in ((1152, 384, "qkv", 0, 0), (384, 384, "proj", 0, 1), (1536, 384, "fc1", 1, 0), (384, 1536, "fc2", 0, 1)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None; y = torch.empty(M, N, device="cuda")
    ref = torch.nn.functional.linear(x[:256].double(), w.double(), b.double())
    if act: ref = torch.nn.functional.gelu(ref)
    if res: ref = ref + r[:256].double()
    ts = {p_: [] for p_, _ in libs}
    errs = {}
    for rd in range(12):                      # rounds INTERLEAVED across the builds: no build owns the cold / ramping clock
        for p_, lib in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): assert lib.tt_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), None, M, N, K, act, 0, st) == 0
            e1.record(); torch.cuda.synchronize()
            if rd >= 3: ts[p_].append(e0.elapsed_time(e1) * 1e2)
            errs[p_] = ((y[:256].double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{name:5s} N={N:5d} K={K:5d}: " + " | ".join(f"{os.path.basename(p_)[3:-3]} {statistics.median(ts[p_]):7.1f} us (err {errs[p_]:.1e})" for p_, _ in libs), flush=True)
