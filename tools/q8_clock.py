#!/usr/bin/env python3
"""The clock the chip holds under gemm_pairs8 (tools/build_variant.sh q8clock gemm_pairs8.hip -DTT_Q8_CLOCK): s_memtime / s_memrealtime around the
whole kernel, printed by the last launch of ~1.5 s of back-to-back launches on random operands (MI355X_MICROARCH.md 'DVFS give-back' item 6)."""
import ctypes as C, os, sys, time, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libq8clock.so"))
lib.tt_linear_fwd_pairs.restype = C.c_int
lib.tt_linear_fwd_pairs.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp, vp]   # ABI 7: + K-split workspace, range flag
lib.tt_split_pairs.restype = C.c_int
lib.tt_split_pairs.argtypes = [vp, vp, ll, vp, vp]
lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
st = torch.cuda.current_stream().cuda_stream
def split(x):
    out = torch.empty((x.shape[0], 2 * x.shape[1]), device="cuda", dtype=torch.float16)
    assert lib.tt_split_pairs(x.data_ptr(), out.data_ptr(), x.numel(), None, st) == 0
    return out
for M, N, K, name in ((25216, 1152, 384, "ViT-S/16 qkv"), (25216, 1536, 384, "ViT-S/16 fc1 (fp32 out)"), (25216, 384, 1536, "ViT-S/16 fc2 (no residual)"), (25216, 2304, 768, "ViT-B/16 qkv")):
    x = split(torch.randn(M, K, device="cuda")); w = split(torch.randn(N, K, device="cuda") * 0.05); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    def go():
        assert lib.tt_linear_fwd_pairs(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, None, M, N, K, 0, _ks(lib, st)[1], _ks(lib, st)[2], None, st) == 0
    for dbg in ([0, 16] if len(sys.argv) < 2 else [int(v) for v in sys.argv[1:]]):   # 16 (needs -DTT_Q8_ABLATE too): two 16x16x32 MFMAs per 32x32x16 one (wrong numbers, same cycles and flops)
        os.environ["TT_Q8_DBG"] = str(dbg)
        print(f"-- TT_Q8_DBG={dbg}  (0 full, 16: 2 x 16x16x32 per 32x32x16 MFMA, 1 no MFMAs, 2 no LDS-DMA, 8 no epilogue, sums combine)")
        lib.tt_set_tuning_knob(b"TT_Q8_ORDER", 3)
        t0 = time.time(); n = 0
        while time.time() - t0 < 1.5:
            for _ in range(200): go()
            torch.cuda.synchronize(); n += 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        print(f"== {name}: {us:.1f} us per launch ({2.0 * M * N * K * 3 / us / 1e6:.0f} TF/s raw)", flush=True)
        lib.tt_set_tuning_knob(b"TT_Q8_ORDER", 103)
        go(); torch.cuda.synchronize(); sys.stdout.flush()
