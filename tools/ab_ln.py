#!/usr/bin/env python3
"""A/B of tt_layernorm_fwd between library builds (interleaved rounds): usage ab_ln.py libA.so libB.so ..."""
import ctypes as C, os, statistics, sys, torch
vp, i32 = C.c_void_p, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p)); lib.tt_layernorm_fwd.restype = C.c_int
    lib.tt_layernorm_fwd.argtypes = [vp] * 6 + [i32, i32, C.c_float, i32, vp]; return lib
libs = [(p, load(p)) for p in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
for rows, D, skip in ((25216, 384, 0), (25088, 384, 197), (25216, 768, 0), (6304, 384, 0)):
    F = rows // (skip - 1) if skip else 0
    x = torch.randn(F * skip if skip else rows, D, device="cuda"); g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
    y = torch.empty(rows, D, device="cuda")
    xin = x.view(F, skip, D)[:, 1:].reshape(-1, D) if skip else x
    ref = torch.nn.functional.layer_norm(xin.double(), (D,), g.double(), b.double(), 1e-6)
    ts = {p: [] for p, _ in libs}; errs = {}
    for rd in range(12):
        for p, lib in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): assert lib.tt_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, rows, D, 1e-6, skip, st) == 0
            e1.record(); torch.cuda.synchronize()
            if rd >= 3: ts[p].append(e0.elapsed_time(e1) * 1e3 / 20)
            errs[p] = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"rows={rows} D={D} skip={skip}: " + " | ".join(f"{os.path.basename(p)[3:-3]} {statistics.median(ts[p]):6.2f} us (err {errs[p]:.1e})" for p, _ in libs), flush=True)
