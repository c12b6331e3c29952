#!/usr/bin/env python3
"""A/B of tt_linear_fwd_planes between library builds (tools/build_variant.sh): usage ab_planes.py libA.so libB.so ...
Times the four ViT-B/16 block shapes (planes 1, bf16 outputs where the step has them) and the four ViT-S/16 shapes (planes 3)."""
import ctypes as C, os, statistics, sys, torch
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p)); lib.tt_linear_fwd_planes.restype = C.c_int
    lib.tt_linear_fwd_planes.argtypes = [vp, ll, vp, ll, i32, vp, vp, vp, vp, vp, ll, i32, i32, i32, i32, i32, vp]; return lib
libs = [(p, load(p)) for p in sys.argv[1:]]
M = 25216
st = torch.cuda.current_stream().cuda_stream
for P, shapes in ((1, [(2304, 768, "qkv", 0, 1, 0), (768, 768, "proj", 0, 0, 1), (3072, 768, "fc1", 1, 1, 0), (768, 3072, "fc2", 0, 0, 1)]),
                  (3, [(1152, 384, "qkv", 0, 0, 0), (384, 384, "proj", 0, 0, 1), (1536, 384, "fc1", 1, 3, 0), (384, 1536, "fc2", 0, 0, 1)])):
    for N, K, name, act, po, res in shapes:
        x = torch.randn(P, M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(P, N, K, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if res else None
        y = torch.empty(M, N, device="cuda") if not po or res else None
        yp = torch.empty(po, M, N, device="cuda", dtype=torch.bfloat16) if po else None
        if res: y = r
        ts = {p_: [] for p_, _ in libs}
        for rd in range(10):                      # rounds INTERLEAVED across the builds (the first ~100 ms of a burst run ~9 % slow)
            for p_, lib in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    assert lib.tt_linear_fwd_planes(x.data_ptr(), M * K, w.data_ptr(), N * K, P, b.data_ptr(), r.data_ptr() if res else None,
                                                    y.data_ptr() if y is not None else None, None, yp.data_ptr() if po else None, M * N, po, M, N, K, act, st) == 0
                e1.record(); torch.cuda.synchronize()
                if rd >= 3: ts[p_].append(e0.elapsed_time(e1) * 1e2)
        print(f"P={P} {name:5s} N={N:5d} K={K:5d}: " + " | ".join(f"{os.path.basename(p_)[3:-3]} {statistics.median(ts[p_]):7.1f} us" for p_, _ in libs), flush=True)
