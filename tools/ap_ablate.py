#!/usr/bin/env python3
"""Where the resident pair attention kernel's time goes: crippled builds (tools/build_variant.sh apN attention_pairs.hip -DTT_AP_DBG=N; bits:
1 no score MFMAs, 2 no exp / split, 4 no P V MFMAs, 8 no V reads, 16 no K reads, 32 no DMA) timed interleaved in one process."""
import ctypes as C, os, statistics, sys, torch
vp, i32 = C.c_void_p, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p)); lib.tt_attention_fwd_pairs.restype = C.c_int
    lib.tt_attention_fwd_pairs.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, vp]; return lib
names = {0: "full", 1: "no score MFMAs", 2: "no exp / split", 4: "no PV MFMAs", 8: "no V reads", 16: "no K reads", 32: "no DMA", 5: "no MFMAs", 24: "no K / V reads",
         7: "no MFMAs, no exp", 31: "DMA + skeleton only", 63: "skeleton only"}
libs = [(d, load(f"tools/bin/libap{d}.so")) for d in names]
F, N, H = 128, 197, 6
qkvp = (torch.randn(F, N, 6 * H * 64, device="cuda") * 0.5).half(); out = torch.empty(F, N, 2 * H * 64, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
res = {d: [] for d, _ in libs}
for rd in range(10):
    for d, lib in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): assert lib.tt_attention_fwd_pairs(qkvp.data_ptr(), out.data_ptr(), None, None, F, N, H, 64, 0.125, st) == 0
        e1.record(); torch.cuda.synchronize()
        if rd >= 2: res[d].append(e0.elapsed_time(e1) * 1e3 / 5)
for d, v in res.items(): print(f"{names[d]:28s} {statistics.median(v):7.1f} us")
