#!/usr/bin/env python3
"""Where attention_fwd_bf16's time goes: interleaved timing of its crippled instantiations (tools/build_variant.sh abfablate
attention_bf16.hip -DTT_ABF_ABLATE; TT_ABF_DBG bit mask: 1 no MFMAs, 2 no K/V LDS-DMA, 4 no Q loads, 8 no exponentials, 16 no stores)
on BASELINE C4's layer (128 frames x 12 heads x 197 tokens)."""
import ctypes as C, os, statistics, sys, torch
vp, i32 = C.c_void_p, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libabfablate.so"))
lib.tt_attention_fwd_bf16.restype = C.c_int
lib.tt_attention_fwd_bf16.argtypes = [vp, vp, i32, i32, i32, i32, C.c_float, vp]
st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "full", 1: "no MFMA", 2: "no K/V DMA", 4: "no Q loads", 8: "no exp", 16: "no stores", 22: "no global traffic at all", 9: "no MFMA, no exp",
         30: "MFMAs + LDS reads + barrier only", 31: "LDS reads + VALU rest"}
F, N, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 197, 12)))
qkv = (torch.randn(F, N, 3 * H * 64, device="cuda")).to(torch.bfloat16)
out = torch.empty(F, N, H * 64, device="cuda", dtype=torch.bfloat16)
ts = {d: [] for d in NAMES}
for rd in range(8):
    for d in NAMES:
        os.environ["TT_ABF_DBG"] = str(d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rc = lib.tt_attention_fwd_bf16(qkv.data_ptr(), out.data_ptr(), F, N, H, 64, 0.125, st)
            assert rc == 0, rc
        e1.record(); torch.cuda.synchronize()
        if rd >= 2: ts[d].append(e0.elapsed_time(e1) * 1e2)
flops = 4.0 * N * N * 64 * F * H
byts = F * N * H * 64 * 2 * 4
for d in NAMES:
    t = statistics.median(ts[d])
    print(f"{NAMES[d]:36s} {t:7.1f} us" + (f"   ({flops / t / 1e6:5.0f} TFLOP/s useful, {byts / t / 1e3:5.0f} GB/s of qkv + out)" if d == 0 else ""), flush=True)
