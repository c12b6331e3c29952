#!/usr/bin/env python3
"""A/B of tt_label_propagate between library builds (interleaved rounds) on the C2 and C4 shapes: usage ab_lp.py libA.so libB.so ..."""
import ctypes as C, os, statistics, sys, torch
vp, i32 = C.c_void_p, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p))
    lib.tt_label_propagate.restype = C.c_int
    lib.tt_label_propagate.argtypes = [vp] * 4 + [i32] * 8 + [C.c_float, i32, vp, C.c_size_t, vp]
    lib.tt_label_propagate_workspace_bytes.restype = C.c_size_t
    lib.tt_label_propagate_workspace_bytes.argtypes = [i32] * 6
    return lib
libs = [(p, load(p)) for p in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
for bs, fs, g, D, K in ((32, 4, 14, 384, 200), (16, 8, 14, 768, 400)):
    n = g * g
    xn = torch.nn.functional.normalize(torch.randn(fs, bs, n, D, device="cuda"), dim=-1)
    seg0 = torch.softmax(torch.randn(bs, n, K, device="cuda"), -1)
    labels = torch.empty(bs, n, dtype=torch.int64, device="cuda")
    nb = libs[0][1].tt_label_propagate_workspace_bytes(bs, fs, g, D, K, 7)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    ts = {p: [] for p, _ in libs}; outs = {}
    for rd in range(10):
        for p, lib in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                assert lib.tt_label_propagate(xn.data_ptr(), seg0.data_ptr(), labels.data_ptr(), None, bs, fs, g, D, K, 7, 6, 5, 0.1, 0, ws.data_ptr(), nb, st) == 0
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: ts[p].append(e0.elapsed_time(e1) * 1e3 / 5)
            outs[p] = labels.clone()
    same = all(torch.equal(outs[libs[0][0]], o) for o in outs.values())
    print(f"bs={bs} fs={fs} D={D} K={K}: " + " | ".join(f"{os.path.basename(p)[3:-3]} {statistics.median(ts[p]):7.1f} us" for p, _ in libs) + f"  labels equal: {same}", flush=True)
