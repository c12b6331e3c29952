#!/usr/bin/env python3
"""A/B of tt_attention_fwd_pairs between library builds in one process, outputs compared bit for bit.  usage: ab_attn_pairs.py libA.so libB.so ..."""
import ctypes as C, os, statistics, sys, torch
vp, i32 = C.c_void_p, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p)); lib.tt_attention_fwd_pairs.restype = C.c_int
    lib.tt_attention_fwd_pairs.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, vp]; return lib
def parse(arg):   # `label=path.so:KNOB=v,...`: the knobs are set before each launch of that entry (the same path may appear twice)
    label, rest = arg.split("=", 1) if ("=" in arg.split(":")[0]) else (None, arg)
    path, _, kn = rest.partition(":")
    knobs = [(k.split("=")[0].encode(), int(k.split("=")[1])) for k in kn.split(",") if k]
    lib = load(path)
    lib.tt_set_tuning_knob.restype = C.c_int; lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
    class L:
        def tt_attention_fwd_pairs(self, *a):
            for k, v in knobs: assert lib.tt_set_tuning_knob(k, v) == 0, k
            return lib.tt_attention_fwd_pairs(*a)
    return (label or os.path.basename(path)), (L() if knobs else lib)
libs = [parse(p) for p in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
for F, N, H, allout in [(128, 197, 6, 0), (128, 197, 6, 1), (128, 197, 12, 0), (32, 197, 6, 1), (64, 256, 6, 0), (3, 50, 2, 1),
                        (64, 785, 6, 0), (16, 785, 6, 1), (128, 785, 6, 0), (5, 300, 3, 1), (64, 785, 12, 0)]:   # (> 256 tokens: the KV-tiled kernel)
    qkvp = (torch.randn(F, N, 6 * H * 64, device="cuda") * 0.5).half()
    out = torch.empty(F, N, 2 * H * 64, device="cuda", dtype=torch.float16)
    o32 = torch.empty(F, N, H * 64, device="cuda") if allout else None
    lse = torch.empty(F, H, N, device="cuda") if allout else None
    def go(lib): assert lib.tt_attention_fwd_pairs(qkvp.data_ptr(), out.data_ptr(), o32.data_ptr() if allout else None, lse.data_ptr() if allout else None, F, N, H, 64, 0.125, st) == 0
    res = {n: [] for n, _ in libs}
    for rd in range(10):
        for n, lib in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): go(lib)
            e1.record(); torch.cuda.synchronize()
            if rd >= 2: res[n].append(e0.elapsed_time(e1) * 1e3 / 5)
    outs = {}
    for n, lib in libs:
        reps = []
        for _ in range(3):
            out.zero_(); go(lib); torch.cuda.synchronize(); reps.append((out.clone(), o32.clone() if allout else None, lse.clone() if allout else None))
        assert all(torch.equal(reps[0][0], q[0]) for q in reps), f"{n}: run-to-run difference"
        outs[n] = reps[0]
    same = all(all((a is None) or torch.equal(a, b) for a, b in zip(outs[n], outs[libs[0][0]])) for n, _ in libs)
    print(f"F={F} N={N} H={H} {'all outputs' if allout else 'pairs out  '}: " + " | ".join(f"{n} {statistics.median(v):6.1f}" for n, v in res.items()) + f"  us  bits {'equal' if same else 'DIFFER'}", flush=True)
