#!/usr/bin/env python3
"""Yardstick, not a dependency: times tt_linear_fwd next to the vendor fp32 GEMM (torch.nn.functional.linear ->
hipBLASLt / rocBLAS sgemm) on the step's forward shapes.  The vendor call does NOT fuse GELU / the residual add, so
its line is the bare GEMM (+bias) only; ours includes the fused epilogue."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from timetuning_amd import hip_ops as ops

SHAPES = [("qkv", 25216, 1152, 384, 0), ("proj", 25216, 384, 384, 0), ("fc1", 25216, 1536, 384, 1), ("fc2", 25216, 384, 1536, 0),
          ("head2", 6272, 1024, 1024, 1)]


def timed_pair(fa, fb, reps=5, rounds=12):
    """Interleaved rounds (clock / thermal state is shared), median of each."""
    ta, tb = [], []
    for rd in range(rounds):
        for fn, acc in ((fa, ta), (fb, tb)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rd >= 2:
                acc.append(e0.elapsed_time(e1) * 1e-3 / reps)
    return statistics.median(ta), statistics.median(tb)


def main():
    torch.manual_seed(0)
    torch.backends.cuda.matmul.allow_tf32 = False
    for name, M, N, K, act in SHAPES:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.02
        b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda")
        t_ours, t_vendor = timed_pair(lambda: ops.linear_fwd(x, w, b, act=act, out=y), lambda: torch.nn.functional.linear(x, w, b))
        ref = torch.nn.functional.linear(x, w, b)
        if act:
            ref = torch.nn.functional.gelu(ref)
        err = ((y - ref).norm() / ref.norm()).item()
        fl = 2.0 * M * N * K
        print(f"{name:6s} M={M} N={N} K={K}: ours {fl / t_ours / 1e12:6.1f} TF ({t_ours * 1e6:7.1f} us)   vendor sgemm {fl / t_vendor / 1e12:6.1f} TF "
              f"({t_vendor * 1e6:7.1f} us)   rel diff {err:.2e}")


if __name__ == "__main__":
    main()
