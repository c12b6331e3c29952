#!/usr/bin/env python3
"""Where a wave's K-tile iteration of gemm_pairs8s_kernel goes (tools/build_variant.sh q8sstamp gemm_pairs8.hip -DTT_Q8S_STAMP): s_memtime
stamps around the top of the iteration (wait for the fragment reads | counted wait for the ring | barrier) and its body (24 MFMAs with the
reads and DMA pieces between them), and around the epilogue; printed by the last of a burst of back-to-back launches.  The stamps (an SMEM
round trip each) perturb the kernel a little: read the SHARES."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ksws import ksplit_ws
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libq8sstamp.so"))
def knob(name, value):
    lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
    assert lib.tt_set_tuning_knob(name.encode(), int(value)) == 0
lib.tt_linear_fwd_pairs.restype = C.c_int
lib.tt_linear_fwd_pairs.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp, vp]
lib.tt_split_pairs.restype = C.c_int
lib.tt_split_pairs.argtypes = [vp, vp, ll, vp, vp]
st = torch.cuda.current_stream().cuda_stream
KS = ksplit_ws(lib, st)
def split(x):
    out = torch.empty((x.shape[0], 2 * x.shape[1]), device="cuda", dtype=torch.float16)
    assert lib.tt_split_pairs(x.data_ptr(), out.data_ptr(), x.numel(), None, st) == 0
    return out
for M, N, K, po, name in ((25216, 1152, 384, 0, "ViT-S/16 qkv (fp32 out)"), (25216, 1536, 384, 1, "ViT-S/16 fc1 (GELU -> pairs)"), (25216, 384, 1536, 0, "ViT-S/16 fc2 (no residual)"),
                          (16384, 1024, 384, 0, "ideal 2 tiles / CU")):
    x = split(torch.randn(M, K, device="cuda")); w = split(torch.randn(N, K, device="cuda") * 0.05); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda") if not po else None
    yp = torch.empty(M, 2 * N, device="cuda", dtype=torch.float16) if po else None
    print(f"== {name}", flush=True)
    def go():
        assert lib.tt_linear_fwd_pairs(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr() if y is not None else None, None, yp.data_ptr() if po else None,
                                       M, N, K, po, KS[1], KS[2], None, st) == 0
    knob("TT_Q8_ORDER", 3)
    for _ in range(800): go()
    knob("TT_Q8_ORDER", 103)
    go()
    torch.cuda.synchronize()
    sys.stdout.flush()
