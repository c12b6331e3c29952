#!/usr/bin/env python3
"""Where a wave's (frame, head) item of attention_fwd_pairs_kernel<7> goes (tools/build_variant.sh apstamp attention_pairs.hip -DTT_AP_STAMP):
s_memtime stamps at the phase boundaries, averaged over the workgroups, per wave and item.  Columns (100 MHz s_memtime ticks are converted
with the measured kernel time): wait+barrier 1 | V issue, Q loads, scores | wait+barrier 2 | K issue, max | exp/split + P V | normalise, stores."""
import ctypes as C, os, sys, torch
vp, i32 = C.c_void_p, C.c_int
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libapstamp.so"))
lib.tt_attention_fwd_pairs.restype = C.c_int
lib.tt_attention_fwd_pairs.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, vp]
st = torch.cuda.current_stream().cuda_stream
F, N, H = 128, 197, 6
qkvp = (torch.randn(F, N, 6 * H * 64, device="cuda") * 0.5).half()
out = torch.empty(F, N, 2 * H * 64, device="cuda", dtype=torch.float16)
G = min(F * H, torch.cuda.get_device_properties(0).multi_processor_count)
stamps = torch.zeros(G * 8 * 4 * 8, device="cuda", dtype=torch.int64)
assert stamps.numel() * 8 <= F * N * H * 64 * 4
buf = torch.zeros(F * N * H * 64, device="cuda")
def go(): assert lib.tt_attention_fwd_pairs(qkvp.data_ptr(), out.data_ptr(), buf.data_ptr(), None, F, N, H, 64, 0.125, st) == 0
for _ in range(20): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
s = buf.view(torch.int64)[: G * 8 * 4 * 8].view(G, 8, 4, 8).cpu().double()
span = (s[:, :, 2, 6].max() - s[:, :, 0, 0].min()).item()
print(f"kernel {us:.1f} us (with launch), first stamp -> last stamp {span:.0f} ticks")
names = ["wait1", "scores", "wait2", "K+max", "PV", "epi", "item"]
for it in range(3):
    print(f"item {it}:   " + " ".join(f"{n:>7s}" for n in names))
    for w in range(8):
        d = [(s[:, w, it, i + 1] - s[:, w, it, i]).mean().item() for i in range(6)] + [(s[:, w, it, 6] - s[:, w, it, 0]).mean().item()]
        print(f"  wave {w}: " + " ".join(f"{v:7.0f}" for v in d))
