#!/usr/bin/env python3
"""Where a wave's steady-state phase of gemm_pairs8 goes (tools/build_variant.sh q8stamp gemm_pairs8.hip -DTT_Q8_STAMP): s_memtime stamps around the
two halves of the load part, the counted wait, the two barriers, the fragment-read wait and the MFMA part, averaged over the steady phases of
one workgroup; printed by the last of a burst of back-to-back launches.  The stamps slow the kernel: the split is indicative."""
import ctypes as C, os, sys, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libq8stamp.so"))
def knob(name, value):   # the library reads its tuning knobs once: flip them through its setter
    lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
    assert lib.tt_set_tuning_knob(name.encode(), int(value)) == 0
lib.tt_linear_fwd_pairs.restype = C.c_int
lib.tt_linear_fwd_pairs.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp, vp]   # ABI 7: + K-split workspace, range flag
lib.tt_split_pairs.restype = C.c_int
lib.tt_split_pairs.argtypes = [vp, vp, ll, vp, vp]
st = torch.cuda.current_stream().cuda_stream
def split(x):
    out = torch.empty((x.shape[0], 2 * x.shape[1]), device="cuda", dtype=torch.float16)
    assert lib.tt_split_pairs(x.data_ptr(), out.data_ptr(), x.numel(), None, st) == 0
    return out
for M, N, K, name in ((25216, 1152, 384, "ViT-S/16 qkv"), (25216, 384, 1536, "ViT-S/16 fc2 (no residual)"), (25216, 2304, 768, "ViT-B/16 qkv")):
    x = split(torch.randn(M, K, device="cuda")); w = split(torch.randn(N, K, device="cuda") * 0.05); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    for order in (3, 0, 1):
        print(f"== {name}, order mode {order}", flush=True)
        def go():
            assert lib.tt_linear_fwd_pairs(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, None, M, N, K, 0, _ks(lib, st)[1], _ks(lib, st)[2], None, st) == 0
        knob("TT_Q8_ORDER", order)
        for _ in range(1500): go()
        knob("TT_Q8_ORDER", 100 + order)
        go()
        torch.cuda.synchronize()
        sys.stdout.flush()
