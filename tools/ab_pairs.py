#!/usr/bin/env python3
"""A/B of tt_linear_fwd_pairs between library builds in one process (tools/build_variant.sh) on the ViT-S/16 and ViT-B/16 block shapes;
outputs compared bit for bit.  usage: ab_pairs.py libA.so libB.so ...
An entry may carry tuning knobs, set before each of its launches: `label=path.so:TT_Q8_STREAM=0,TT_Q8_KSPLIT=1` (A/B of two dispatch settings
of ONE library: the same path may appear twice)."""
import ctypes as C, os, statistics, sys, torch
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _ksws import ksplit_ws
_KS = {}
def _ks(lib, st):
    lib = getattr(lib, "raw", lib)
    if id(lib) not in _KS: _KS[id(lib)] = ksplit_ws(lib, st)
    return _KS[id(lib)]
vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
def load(p):
    lib = C.CDLL(os.path.abspath(p))
    lib.tt_linear_fwd_pairs.restype = C.c_int
    lib.tt_linear_fwd_pairs.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp, vp]   # ABI 7: + K-split workspace, range flag
    lib.tt_split_pairs.restype = C.c_int
    lib.tt_split_pairs.argtypes = [vp, vp, ll, vp, vp]
    return lib
def parse(arg):
    label, rest = arg.split("=", 1) if ("=" in arg.split(":")[0]) else (None, arg)
    path, _, kn = rest.partition(":")
    knobs = [(k.split("=")[0].encode(), int(k.split("=")[1])) for k in kn.split(",") if k]
    lib = load(path)
    lib.tt_set_tuning_knob.restype = C.c_int
    lib.tt_set_tuning_knob.argtypes = [C.c_char_p, C.c_int]
    class L:   # a library handle that applies its knobs before every call
        raw = lib
        def __getattr__(self, name):
            fn = getattr(lib, name)
            def call(*a):
                for k, v in knobs: assert lib.tt_set_tuning_knob(k, v) == 0, k
                return fn(*a)
            return call if name in ("tt_linear_fwd_pairs",) else fn
    return (label or os.path.basename(path) + (":" + kn if kn else "")), (L() if knobs else lib)
libs = [parse(p) for p in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
def split(x):
    out = torch.empty((x.shape[0], 2 * x.shape[1]), device="cuda", dtype=torch.float16)
    assert libs[0][1].tt_split_pairs(x.data_ptr(), out.data_ptr(), x.numel(), None, st) == 0
    return out
cases = [(25216, 1152, 384, 0, 0, 0, "qkv"), (25216, 384, 384, 0, 0, 1, "proj"), (25216, 1536, 384, 1, 1, 0, "fc1"), (25216, 384, 1536, 0, 0, 1, "fc2"),
         (25216, 2304, 768, 0, 0, 0, "B qkv"), (25216, 768, 768, 0, 0, 1, "B proj"), (25216, 3072, 768, 1, 1, 0, "B fc1"), (25216, 768, 3072, 0, 0, 1, "B fc2"),
         (6304, 1536, 384, 1, 1, 0, "kept fc1"), (6500, 384, 384, 0, 0, 1, "ragged"),
         # the projection head on 6272 rows (K % 96 != 0: the round-4 persistent kernel refused them)
         (6272, 1024, 384, 1, 1, 0, "head 1"), (6272, 1024, 1024, 1, 1, 0, "head 2"), (6272, 512, 1024, 1, 1, 0, "head 3"), (6272, 256, 512, 0, 0, 0, "head 4"),
         # kept-frame launches of the trainable blocks (6304 rows: 75 tiles at N = 384)
         (6304, 384, 384, 0, 0, 1, "kept proj"), (6304, 384, 1536, 0, 0, 1, "kept fc2"), (6304, 1152, 384, 0, 1, 0, "kept qkv"),
         (18912, 384, 384, 0, 0, 1, "rest proj"), (18912, 1536, 384, 1, 1, 0, "rest fc1"),
         # C1 (2 clips x 2 frames = 788 rows): launch-latency regime of the general kernel
         (788, 1152, 384, 0, 1, 0, "C1 qkv"), (788, 384, 384, 0, 0, 1, "C1 proj"), (788, 1536, 384, 1, 1, 0, "C1 fc1"), (788, 384, 1536, 0, 0, 1, "C1 fc2")]
if os.environ.get("TT_AB_CASES") == "exact":   # whole rounds for both tile geometries (256 x 128 on 256 CUs, 128 x 128 on 512 workgroups): the main loop without quantisation
    cases = [(32768, 1024, 384, 0, 0, 0, "x f32 K384"), (32768, 1024, 384, 0, 1, 0, "x pair K384"), (32768, 1024, 384, 1, 1, 0, "x gelu K384"), (32768, 1024, 1536, 0, 0, 1, "x res K1536"),
             (32768, 1024, 768, 0, 1, 0, "x pair K768")]
if os.environ.get("TT_AB_CASES") == "small":   # the C2 step's launches below 96 256 x 128 tiles (general kernel; knob TT_Q4_SMALL: the four-wave kernel)
    cases = [(6304, 384, 384, 0, 0, 1, "kept proj"), (6304, 384, 1536, 0, 0, 1, "kept fc2"), (6304, 384, 384, 0, 0, 0, "dgrad proj"), (6304, 384, 1152, 0, 0, 0, "dgrad qkv"),
             (6304, 384, 1536, 0, 0, 0, "dgrad fc1"), (6272, 384, 1024, 0, 0, 0, "dgrad hd1"), (6272, 256, 512, 0, 0, 0, "head 4"), (6272, 256, 512, 0, 1, 0, "head 4 pr"),
             (12608, 384, 384, 0, 0, 1, "half proj"), (12608, 384, 1536, 0, 0, 1, "half fc2")]
tot = {n: 0.0 for n, _ in libs}
for M, N, K, act, po, res, name in cases:
    x = split(torch.randn(M, K, device="cuda")); w = split(torch.randn(N, K, device="cuda") * 0.05)
    b = torch.randn(N, device="cuda"); r0 = torch.randn(M, N, device="cuda") if res else None
    r = r0.clone() if res else None
    y = r if res else (torch.empty(M, N, device="cuda") if not po else None)
    yp = torch.empty(M, 2 * N, device="cuda", dtype=torch.float16) if po else None
    def go(lib):
        rc = lib.tt_linear_fwd_pairs(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr() if y is not None else None, None,
                                     yp.data_ptr() if po else None, M, N, K, act, _ks(lib, st)[1], _ks(lib, st)[2], None, st)
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
        reps = []
        for _ in range(3):
            if res: r.copy_(r0)
            go(lib); torch.cuda.synchronize(); reps.append((y if y is not None else yp).clone())
        assert all(torch.equal(reps[0], q) for q in reps), f"{n}: run-to-run difference on {name}"
        outs[n] = reps[0]
    same = all(torch.equal(outs[n], outs[libs[0][0]]) for n, _ in libs)
    for n, _ in libs:
        if name in ("qkv", "proj", "fc1", "fc2"): tot[n] += statistics.median(ts[n])
    print(f"{name:9s} M={M} N={N} K={K}: " + " | ".join(f"{n} {statistics.median(ts[n]):7.1f}" for n, _ in libs) + f"  us   bits {'equal' if same else 'DIFFER'}", flush=True)
print("ViT-S/16 block (qkv + proj + fc1 + fc2): " + " | ".join(f"{n} {v:7.1f}" for n, v in tot.items()) + "  us")
