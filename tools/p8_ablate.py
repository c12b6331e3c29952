#!/usr/bin/env python3
"""Where gemm_planes8's time goes: interleaved timing of its crippled instantiations (tools/build_variant.sh p8ablate gemm_planes8.hip
-DTT_P8_ABLATE; TT_P8_DBG bit mask: 1 no MFMAs, 2 no LDS-DMA, 4 no fragment reads, 8 no epilogue, 16 epilogue without global traffic)
on shapes with an exact tile count per CU (16384 x 2048) and on the ViT-B/16 / ViT-S/16 block shapes."""
import ctypes as C, os, statistics, sys, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libp8ablate.so"))
lib.tt_linear_fwd_planes.restype = C.c_int
lib.tt_linear_fwd_planes.argtypes = [vp, ll, vp, ll, i32, vp, vp, vp, vp, vp, ll, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]   # ABI 7: + K-split workspace
lib.tt_split_planes.restype = C.c_int
lib.tt_split_planes.argtypes = [vp, vp, ll, i32, ll, vp]
st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "full", 32: "full, one warm tile", 40: "noEpi, one warm tile", 41: "noEpi+noMFMA, one warm tile", 1: "noMFMA", 2: "noDMA", 4: "noLDSrd", 8: "noEpi", 16: "epiNoGlobal", 9: "noEpi+noMFMA", 11: "noEpi+noMFMA+noDMA", 13: "noEpi+noMFMA+noLDS", 15: "barriers only", 14: "MFMA only(noEpi,noDMA,noLDS)", 64: "reads before DMA", 72: "noEpi+reads before DMA"}
def split(x, P):
    out = torch.empty((P,) + tuple(x.shape), device="cuda", dtype=torch.bfloat16)
    assert lib.tt_split_planes(x.data_ptr(), out.data_ptr(), x.numel(), P, x.numel(), st) == 0
    return out
cases = [(1, 16384, 2048, 768, 0, 1, 0, "ideal-K768 bf16out"), (1, 16384, 2048, 3072, 0, 0, 1, "ideal-K3072 f32+res"),
         (1, 25216, 2304, 768, 0, 1, 0, "qkv"), (1, 25216, 3072, 768, 1, 1, 0, "fc1"), (1, 25216, 768, 3072, 0, 0, 1, "fc2"),
         (3, 16384, 1024, 384, 0, 0, 0, "ideal3-K384 f32out"), (3, 25216, 1152, 384, 0, 0, 0, "qkv3"), (3, 25216, 1536, 384, 1, 3, 0, "fc1_3"),
         (3, 25216, 384, 1536, 0, 0, 1, "fc2_3")]
only = sys.argv[1:] 
for P, M, N, K, act, po, res, name in cases:
    if only and name not in only: continue
    x = split(torch.randn(M, K, device="cuda"), P); w = split(torch.randn(N, K, device="cuda") * 0.05, P)
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if res else None
    y = r if res else (torch.empty(M, N, device="cuda") if not po else None)
    yp = torch.empty(po, M, N, device="cuda", dtype=torch.bfloat16) if po else None
    ts = {d: [] for d in NAMES}
    for rd in range(8):
        for d in NAMES:
            os.environ["TT_P8_DBG"] = str(d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rc = lib.tt_linear_fwd_planes(x.data_ptr(), M * K, w.data_ptr(), N * K, P, b.data_ptr(), r.data_ptr() if res else None,
                                              y.data_ptr() if y is not None else None, None, yp.data_ptr() if po else None, M * N, po, M, N, K, act, _ks(lib, st)[1], _ks(lib, st)[2], st)
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: ts[d].append(e0.elapsed_time(e1) * 1e2)
    nprod = P * (P + 1) // 2
    full = statistics.median(ts[0])
    print(f"P={P} {name:20s} M={M} N={N} K={K}: " + " | ".join(f"{NAMES[d]} {statistics.median(ts[d]):7.1f}" for d in NAMES) +
          f"  us   (full = {2.0 * M * N * K * nprod / full / 1e6:6.0f} TF/s raw)", flush=True)
