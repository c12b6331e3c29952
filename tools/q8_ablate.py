#!/usr/bin/env python3
"""Where gemm_pairs8's time goes: interleaved timing of its crippled instantiations (tools/build_variant.sh q8ablate gemm_pairs8.hip
-DTT_Q8_ABLATE; TT_Q8_DBG bit mask: 1 no MFMAs, 2 no LDS-DMA, 8 no epilogue) on the ViT-S/16 / ViT-B/16 block shapes and on a shape with
an exact tile count per CU."""
import ctypes as C, os, statistics, sys, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libq8ablate.so"))
lib.tt_linear_fwd_pairs.restype = C.c_int
lib.tt_linear_fwd_pairs.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp, vp]   # ABI 7: + K-split workspace, range flag
lib.tt_split_pairs.restype = C.c_int
lib.tt_split_pairs.argtypes = [vp, vp, ll, vp, vp]
st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "full", 1: "noMFMA", 2: "noDMA", 8: "noEpi", 9: "noEpi+noMFMA", 10: "noEpi+noDMA", 3: "noMFMA+noDMA", 11: "reads+barriers only"}
if "hot" in sys.argv:   # round 6: + the hot-operand instantiations of gemm_pairs8s_kernel (every item streams operand tile (0, 0))
    sys.argv.remove("hot")
    NAMES.update({4: "hot", 12: "hot+noEpi", 13: "hot+noEpi+noMFMA"})
def split(x):
    out = torch.empty((x.shape[0], 2 * x.shape[1]), device="cuda", dtype=torch.float16)
    assert lib.tt_split_pairs(x.data_ptr(), out.data_ptr(), x.numel(), None, st) == 0
    return out
# (M, N, K, act, pairs out, residual, name)
cases = [(16384, 1024, 384, 0, 0, 0, "ideal 2 tiles/CU K384 f32out"), (16384, 1024, 1536, 0, 0, 1, "ideal 2 tiles/CU K1536 f32+res"),
         (25216, 1152, 384, 0, 0, 0, "qkv"), (25216, 384, 384, 0, 0, 1, "proj"), (25216, 1536, 384, 1, 1, 0, "fc1"), (25216, 384, 1536, 0, 0, 1, "fc2"),
         (25216, 2304, 768, 0, 0, 0, "B qkv"), (25216, 768, 3072, 0, 0, 1, "B fc2")]
# first argument `stream=0|1`: which persistent kernel (TT_Q8_STREAM: 1 = gemm_pairs8s_kernel, round 5; 0 = gemm_pairs8_kernel)
args = sys.argv[1:]
if args and args[0].startswith("stream="):
    lib.tt_set_tuning_knob.restype = C.c_int
    lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
    assert lib.tt_set_tuning_knob(b"TT_Q8_STREAM", int(args[0][7:])) == 0
    print(f"== TT_Q8_STREAM = {args[0][7:]}", flush=True)
    args = args[1:]
only = args
for M, N, K, act, po, res, name in cases:
    if only and name not in only: continue
    x = split(torch.randn(M, K, device="cuda")); w = split(torch.randn(N, K, device="cuda") * 0.05)
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if res else None
    y = r if res else (torch.empty(M, N, device="cuda") if not po else None)
    yp = torch.empty(M, 2 * N, device="cuda", dtype=torch.float16) if po else None
    ts = {d: [] for d in NAMES}
    for rd in range(8):
        for d in NAMES:
            os.environ["TT_Q8_DBG"] = str(d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rc = lib.tt_linear_fwd_pairs(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr() if y is not None else None,
                                             None, yp.data_ptr() if po else None, M, N, K, act, _ks(lib, st)[1], _ks(lib, st)[2], None, st)
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: ts[d].append(e0.elapsed_time(e1) * 1e2)
    full = statistics.median(ts[0])
    print(f"{name:32s} M={M} N={N} K={K}: " + " | ".join(f"{NAMES[d]} {statistics.median(ts[d]):7.1f}" for d in NAMES) +
          f"  us   (full = {2.0 * M * N * K * 3 / full / 1e6:6.0f} TF/s raw, {2.0 * M * N * K / full / 1e6:5.0f} equivalent)", flush=True)
