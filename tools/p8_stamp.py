#!/usr/bin/env python3
"""Where a wave's steady-state phase of gemm_planes8 goes (tools/build_variant.sh p8stamp gemm_planes8.hip -DTT_P8_STAMP -DTT_P8_CLOCK): s_memtime
stamps around the DMA issue, the fragment reads, the counted wait, the two barriers and the MFMA part, averaged over the steady phases of
one workgroup, for a wave of each group; printed by the last of ~0.3 s of back-to-back launches."""
import ctypes as C, os, sys, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", sys.argv[1] if len(sys.argv) > 1 else "libp8stamp.so"))
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
for P, M, N, K, po, res, name in ((1, 25216, 2304, 768, 1, 0, "ViT-B/16 qkv"), (1, 25216, 768, 3072, 0, 1, "ViT-B/16 fc2"), (3, 25216, 1152, 384, 0, 0, "ViT-S/16 qkv, 3 planes")):
    x = split(torch.randn(M, K, device="cuda"), P); w = split(torch.randn(N, K, device="cuda") * 0.05, P); b = torch.randn(N, device="cuda")
    y = torch.randn(M, N, device="cuda") if not po else None
    yp = torch.empty(po, M, N, device="cuda", dtype=torch.bfloat16) if po else None
    print(f"== P={P} {name}", flush=True)
    def go():
        assert lib.tt_linear_fwd_planes(x.data_ptr(), M * K, w.data_ptr(), N * K, P, b.data_ptr(), y.data_ptr() if res else None, y.data_ptr() if y is not None else None,
                                        None, yp.data_ptr() if po else None, M * N, po, M, N, K, 0, _ks(lib, st)[1], _ks(lib, st)[2], st) == 0
    knob("TT_P8_CLOCK_PRINT", 0)
    for _ in range(3000): go()
    knob("TT_P8_CLOCK_PRINT", 1)
    go()
    torch.cuda.synchronize()
    sys.stdout.flush()
