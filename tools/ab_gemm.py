#!/usr/bin/env python3
"""A/B of tt_linear_fwd between two builds of the library in ONE process (interleaved rounds, median + min).
usage: ab_gemm.py libA.so libB.so ..."""
import ctypes as C
import os
import statistics
import sys

import torch

SHAPES = [("qkv", 25216, 1152, 384, 0), ("proj", 25216, 384, 384, 0), ("fc1", 25216, 1536, 384, 1), ("fc2", 25216, 384, 1536, 0),
          ("head2", 6272, 1024, 1024, 1)]


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.tt_linear_fwd.restype = C.c_int
    lib.tt_linear_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]   # ABI 8: + precision
    return lib


def main():
    libs = [(p, load(p)) for p in sys.argv[1:]]
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    for name, M, N, K, act in SHAPES:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.02
        b = torch.zeros(N, device="cuda")
        y = torch.empty(M, N, device="cuda")
        res = {p: [] for p, _ in libs}
        for rd in range(12):
            for p, lib in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    rc = lib.tt_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, M, N, K, act, 0, st)
                    assert rc == 0
                e1.record()
                torch.cuda.synchronize()
                if rd >= 2:
                    res[p].append(e0.elapsed_time(e1) * 1e-3 / 5)
        outs = []
        for p, lib in libs:  # the builds must agree bit for bit unless the summation order changed
            yo = torch.empty(M, N, device="cuda")
            lib.tt_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, yo.data_ptr(), None, M, N, K, act, 0, st)
            outs.append(yo)
        torch.cuda.synchronize()
        diff = max((o - outs[0]).abs().max().item() for o in outs)
        print(f"  max |diff| vs first build: {diff:.3e}")
        fl = 2.0 * M * N * K
        print(name, M, N, K, " | ".join(f"{os.path.basename(p)}: med {fl / statistics.median(v) / 1e12:6.1f} max {fl / min(v) / 1e12:6.1f} TF" for p, v in res.items()))


if __name__ == "__main__":
    main()
