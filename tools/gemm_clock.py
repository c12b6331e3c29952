#!/usr/bin/env python3
"""In-kernel clock of the dominant GEMM (diagnostic build with -DTT_CLOCK_STAMP, see gemm_nt_fast.hip).

Runs one forward shape back to back on random operands for a few seconds (so that the power management has settled),
then reads the per-workgroup stamps of the last launch: clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, median over
workgroups.  The f32 matrix peak at THAT clock is 256 CUs x 4 SIMDs x 64 flop/cycle x clock.
usage: gemm_clock.py libclock.so [seconds]"""
import ctypes as C
import os
import statistics
import sys
import time

import torch

SHAPES = [("qkv", 25216, 1152, 384, 0), ("fc1", 25216, 1536, 384, 1), ("fc2", 25216, 384, 1536, 0), ("proj", 25216, 384, 384, 0)]


def main():
    lib = C.CDLL(os.path.abspath(sys.argv[1]))
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    lib.tt_linear_fwd.restype = C.c_int
    lib.tt_linear_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]
    lib.tt_debug_read_clock_stamps.argtypes = [C.c_void_p, C.c_int]
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    for zero in (False, True):
        for name, M, N, K, act in SHAPES:
            x = torch.zeros(M, K, device="cuda") if zero else torch.randn(M, K, device="cuda")
            w = torch.zeros(N, K, device="cuda") if zero else torch.randn(N, K, device="cuda") * 0.02
            b = torch.zeros(N, device="cuda")
            y = torch.empty(M, N, device="cuda")
            t0 = time.time()
            n = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            while time.time() - t0 < secs:
                e0.record()
                for _ in range(50):
                    lib.tt_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, M, N, K, act, 0, st)
                e1.record()
                torch.cuda.synchronize()
                n += 50
            t = e0.elapsed_time(e1) * 1e-3 / 50
            tiles = min(8192, (M // 64) * (N // 128))
            buf = (C.c_ulonglong * (6 * tiles))()
            lib.tt_debug_read_clock_stamps(buf, 6 * tiles)
            clk, pro, loop, epi, entry, exit_ = [], [], [], [], [], []
            for i in range(tiles):
                c0, r0, c1, r1, re, rx = buf[6 * i:6 * i + 6]
                if r1 > r0:
                    clk.append((c1 - c0) / (r1 - r0) * 100e6)
                    pro.append((r0 - re) * 0.01); loop.append((r1 - r0) * 0.01); epi.append((rx - r1) * 0.01)   # us (100 MHz ticks)
                    entry.append(re); exit_.append(rx)
            span = (max(exit_) - min(entry)) * 0.01
            # time-line of one launch: how much of the workgroup-resident time is main loop
            print(f"   workgroup phases (us, median): prologue {statistics.median(pro):.2f}  main loop {statistics.median(loop):.2f}  epilogue "
                  f"{statistics.median(epi):.2f};  launch span {span:.1f} us; resident workgroups (mean over the span) "
                  f"{(sum(pro) + sum(loop) + sum(epi)) / span / 256:.2f} per CU, of which in the main loop {sum(loop) / span / 256:.2f}; "
                  f"first entry -> last entry {(max(entry) - min(entry)) * 0.01:.1f} us")
            ghz = statistics.median(clk) / 1e9
            tf = 2.0 * M * N * K / t / 1e12
            peak = 256 * 4 * 64 * ghz / 1e3
            print(f"{'zero  ' if zero else 'random'} {name:5s} M={M} N={N} K={K}: {tf:6.1f} TFLOP/s, in-kernel clock {ghz:.3f} GHz "
                  f"(p10 {sorted(clk)[len(clk) // 10] / 1e9:.3f}, p90 {sorted(clk)[9 * len(clk) // 10] / 1e9:.3f}) -> f32 matrix peak at that clock "
                  f"{peak:6.1f} TFLOP/s, fraction {tf / peak:.3f}; of the 2.4 GHz peak {tf / 157.3:.3f}")


if __name__ == "__main__":
    main()
