#!/usr/bin/env python3
"""A/B of tt_linear_fwd_planes between library builds in one process (tools/build_variant.sh) on the ViT-B/16 (P = 1) and ViT-S/16 (P = 3)
block shapes; outputs compared bit for bit.  usage: ab_planes.py libA.so libB.so ..."""
import ctypes as C, os, statistics, sys, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p))
    lib.tt_linear_fwd_planes.restype = C.c_int
    lib.tt_linear_fwd_planes.argtypes = [vp, ll, vp, ll, i32, vp, vp, vp, vp, vp, ll, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]   # ABI 7: + K-split workspace
    lib.tt_split_planes.restype = C.c_int
    lib.tt_split_planes.argtypes = [vp, vp, ll, i32, ll, vp]
    return lib
libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
def split(x, P):
    out = torch.empty((P,) + tuple(x.shape), device="cuda", dtype=torch.bfloat16)
    assert libs[0][1].tt_split_planes(x.data_ptr(), out.data_ptr(), x.numel(), P, x.numel(), st) == 0
    return out
cases = [(1, 25216, 2304, 768, 0, 1, 0, "qkv"), (1, 25216, 3072, 768, 1, 1, 0, "fc1"), (1, 25216, 768, 768, 0, 0, 1, "proj"), (1, 25216, 768, 3072, 0, 0, 1, "fc2"),
         (3, 25216, 1152, 384, 0, 0, 0, "qkv3"), (3, 25216, 1536, 384, 1, 3, 0, "fc1_3"), (3, 25216, 384, 384, 0, 0, 1, "proj3"), (3, 25216, 384, 1536, 0, 0, 1, "fc2_3")]
for P, M, N, K, act, po, res, name in cases:
    x = split(torch.randn(M, K, device="cuda"), P); w = split(torch.randn(N, K, device="cuda") * 0.05, P)
    b = torch.randn(N, device="cuda"); r0 = torch.randn(M, N, device="cuda") if res else None
    r = r0.clone() if res else None
    y = r if res else (torch.empty(M, N, device="cuda") if not po else None)
    yp = torch.empty(po, M, N, device="cuda", dtype=torch.bfloat16) if po else None
    def go(lib):
        rc = lib.tt_linear_fwd_planes(x.data_ptr(), M * K, w.data_ptr(), N * K, P, b.data_ptr(), r.data_ptr() if res else None,
                                      y.data_ptr() if y is not None else None, None, yp.data_ptr() if po else None, M * N, po, M, N, K, act, _ks(lib, st)[1], _ks(lib, st)[2], st)
        assert rc == 0, rc
    ts = {n: [] for n, _ in libs}
    for rd in range(8):
        for n, lib in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): go(lib)
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: ts[n].append(e0.elapsed_time(e1) * 1e2)
    outs = {}
    for n, lib in libs:
        if res: r.copy_(r0)
        go(lib); torch.cuda.synchronize(); outs[n] = (y if y is not None else yp).clone()
    same = all(torch.equal(outs[n], outs[libs[0][0]]) for n, _ in libs)
    print(f"P={P} {name:6s} M={M} N={N} K={K}: " + " | ".join(f"{n} {statistics.median(ts[n]):7.1f}" for n, _ in libs) + f"  us   bits {'equal' if same else 'DIFFER'}", flush=True)
