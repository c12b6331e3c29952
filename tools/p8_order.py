#!/usr/bin/env python3
"""gemm_planes8's load-part orders (TT_P8_ORDER, read per call) interleaved in one process on the ViT-B/16 (P = 1) and ViT-S/16 (P = 3) block shapes."""
import ctypes as C, os, statistics, sys, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(root, "timetuning_amd", "libtimetuning_hip.so"))
def knob(name, value):   # the library reads its tuning knobs once: flip them through its setter
    lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
    assert lib.tt_set_tuning_knob(name.encode(), int(value)) == 0
lib.tt_linear_fwd_planes.restype = C.c_int
lib.tt_linear_fwd_planes.argtypes = [vp, ll, vp, ll, i32, vp, vp, vp, vp, vp, ll, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]   # ABI 7: + K-split workspace
lib.tt_split_planes.restype = C.c_int
lib.tt_split_planes.argtypes = [vp, vp, ll, i32, ll, vp]
st = torch.cuda.current_stream().cuda_stream
ORDERS = [int(v) for v in os.environ.get("P8_ORDERS", "0 1 2 3").split()]
def split(x, P):
    out = torch.empty((P,) + tuple(x.shape), device="cuda", dtype=torch.bfloat16)
    assert lib.tt_split_planes(x.data_ptr(), out.data_ptr(), x.numel(), P, x.numel(), st) == 0
    return out
cases = [(1, 25216, 2304, 768, 0, 1, 0, "qkv"), (1, 25216, 3072, 768, 1, 1, 0, "fc1"), (1, 25216, 768, 768, 0, 0, 1, "proj"), (1, 25216, 768, 3072, 0, 0, 1, "fc2"),
         (3, 25216, 1152, 384, 0, 0, 0, "qkv3"), (3, 25216, 1536, 384, 1, 3, 0, "fc1_3"), (3, 25216, 384, 384, 0, 0, 1, "proj3"), (3, 25216, 384, 1536, 0, 0, 1, "fc2_3")]
for P, M, N, K, act, po, res, name in cases:
    x = split(torch.randn(M, K, device="cuda"), P); w = split(torch.randn(N, K, device="cuda") * 0.05, P)
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if res else None
    y = r if res else (torch.empty(M, N, device="cuda") if not po else None)
    yp = torch.empty(po, M, N, device="cuda", dtype=torch.bfloat16) if po else None
    ts = {d: [] for d in ORDERS}
    outs = {}
    for rd in range(8):
        for d in ORDERS:
            knob("TT_P8_ORDER", d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rc = lib.tt_linear_fwd_planes(x.data_ptr(), M * K, w.data_ptr(), N * K, P, b.data_ptr(), r.data_ptr() if res else None,
                                              y.data_ptr() if y is not None else None, None, yp.data_ptr() if po else None, M * N, po, M, N, K, act, _ks(lib, st)[1], _ks(lib, st)[2], st)
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: ts[d].append(e0.elapsed_time(e1) * 1e2)
    if not res:   # same bits whatever the order
        for d in ORDERS:
            knob("TT_P8_ORDER", d)
            lib.tt_linear_fwd_planes(x.data_ptr(), M * K, w.data_ptr(), N * K, P, b.data_ptr(), None, y.data_ptr() if y is not None else None, None,
                                     yp.data_ptr() if po else None, M * N, po, M, N, K, act, _ks(lib, st)[1], _ks(lib, st)[2], st)
            torch.cuda.synchronize(); outs[d] = (y if y is not None else yp).clone()
        assert all(torch.equal(outs[d], outs[ORDERS[0]]) for d in ORDERS), "orders disagree"
    print(f"P={P} {name:6s} M={M} N={N} K={K}: " + " | ".join(f"order {d} {statistics.median(ts[d]):7.1f}" for d in ORDERS) + "  us", flush=True)
