#!/usr/bin/env python3
"""In-kernel clock of gemm_planes8 and of its crippled variants (tools/build_variant.sh p8clock gemm_planes8.hip -DTT_P8_ABLATE -DTT_P8_CLOCK):
~1 s of back-to-back launches per variant, the last launch prints s_memtime / s_memrealtime."""
import ctypes as C, os, sys, time, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libp8clock.so"))
def knob(name, value):   # the library reads its tuning knobs once: flip them through its setter
    lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
    assert lib.tt_set_tuning_knob(name.encode(), int(value)) == 0
lib.tt_linear_fwd_planes.restype = C.c_int
lib.tt_linear_fwd_planes.argtypes = [vp, ll, vp, ll, i32, vp, vp, vp, vp, vp, ll, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]   # ABI 7: + K-split workspace
lib.tt_split_planes.restype = C.c_int
lib.tt_split_planes.argtypes = [vp, vp, ll, i32, ll, vp]
st = torch.cuda.current_stream().cuda_stream
def split(x, P):
    out = torch.empty((P,) + tuple(x.shape), device="cuda", dtype=torch.bfloat16)
    assert lib.tt_split_planes(x.data_ptr(), out.data_ptr(), x.numel(), P, x.numel(), st) == 0
    return out
for P, M, N, K, po in ((1, 25216, 2304, 768, 1), (3, 25216, 1152, 384, 0)):
    x = split(torch.randn(M, K, device="cuda"), P); w = split(torch.randn(N, K, device="cuda") * 0.05, P); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda") if not po else None
    yp = torch.empty(po, M, N, device="cuda", dtype=torch.bfloat16) if po else None
    for dbg, name in ((0, "full"), (14, "MFMA only (pseudo-random register operands)"), (9, "skeleton (no MFMA, no epilogue)"), (8, "no epilogue"), (10, "no epilogue, no DMA (MFMA + LDS reads of the random data the full variant left in LDS)"), (12, "no epilogue, no LDS reads (MFMA on pseudo-random registers + DMA)")):
        os.environ["TT_P8_DBG"] = str(dbg)
        print(f"== P={P} {name}", flush=True)
        def go():
            lib.tt_linear_fwd_planes(x.data_ptr(), M * K, w.data_ptr(), N * K, P, b.data_ptr(), None, y.data_ptr() if y is not None else None, None,
                                     yp.data_ptr() if po else None, M * N, po, M, N, K, 0, _ks(lib, st)[1], _ks(lib, st)[2], st)
        knob("TT_P8_CLOCK_PRINT", 0)
        for _ in range(6000): go()      # ~0.6 s of back-to-back launches: the clock has settled
        knob("TT_P8_CLOCK_PRINT", 1)
        go()
        torch.cuda.synchronize()
        sys.stdout.flush()
        print(f"   ^ {name}", flush=True)
