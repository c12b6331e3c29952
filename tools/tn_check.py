#!/usr/bin/env python3
"""gemm_pairs_tn (weight gradient from row pairs, transposing LDS reads) against fp64 and the transposed-operand route, with the time of
each route INCLUDING its preparation passes (dy: split / dual; x: transpose).   python tools/tn_check.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops

bad = 0
for M, N, K in [(6304, 1152, 384), (6304, 384, 384), (6304, 1536, 384), (6304, 384, 1536), (6300, 384, 384), (6272, 256, 384), (1000, 128, 128),
                (37, 128, 256), (12544, 1152, 384), (6304, 2304, 768), (6304, 768, 3072), (3152, 3072, 768)]:
    torch.manual_seed(2)
    dy = torch.randn(M, N, device="cuda") * 0.02; x = torch.randn(M, K, device="cuda")
    ref = dy.double().t() @ x.double()
    xp = ops.split_pairs(x)
    def route_tn():
        _, dyr, db = ops.split_pairs_dual(dy, want_row=True, want_colsum=True, want_t=False)
        return ops.linear_bwd_weight_pairs_tn(dyr, xp), db
    def route_t():
        xT = ops.transpose_pairs(xp)
        dyT, dyr, db = ops.split_pairs_dual(dy, want_row=True, want_colsum=True, rpad=xT.shape[1] // 2)
        lib = ops._lib.load()
        Mpad = xT.shape[1] // 2
        dw = torch.empty((N, K), device="cuda")
        nb = lib.tt_linear_bwd_weight_pairs_workspace_bytes(N, K, Mpad); ws = ops._ws(nb, dy.device)
        ops._lib.check(lib.tt_linear_bwd_weight_pairs(ops._p(dyT), ops._p(xT), ops._p(dw), None, N, K, Mpad, ops._p(ws), nb, ops._stream()), "wgrad")
        return dw, db
    dw, db = route_tn()
    dw2, db2 = route_t()
    dw32, _ = ops.linear_bwd_weight(dy, x, need_bias=False) if hasattr(ops, "linear_bwd_weight") else (None, None)
    e = lambda a: ((a.double() - ref).norm() / ref.norm()).item()
    same = all(torch.equal(route_tn()[0], dw) for _ in range(4))
    edb = ((db.double() - dy.double().sum(0)).norm() / dy.double().sum(0).norm()).item()
    def t(fn, n=10):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
        return statistics.median(ts)
    _, dyr, _ = ops.split_pairs_dual(dy, want_row=True, want_colsum=False, want_t=False)
    t_kernel = t(lambda: ops.linear_bwd_weight_pairs_tn(dyr, xp))
    ok = same and e(dw) < 6e-7 and e(dw) <= 1.05 * e(dw32) and edb < 1e-6
    bad += not ok
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: rel-L2 vs fp64: tn {e(dw):.2e} | transposed route {e(dw2):.2e} | f32 kernel {e(dw32):.2e} | db {edb:.1e} | repeat {'ok' if same else 'DIFFERS'} | "
          f"us: tn route {t(route_tn):7.1f} (kernel + fold {t_kernel:6.1f} = {fl / t_kernel * 1e-6:4.0f} TF-eq) | transposed route {t(route_t):7.1f}" + ("" if ok else "   FAIL"), flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
