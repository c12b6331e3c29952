#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (development aid): the GEMM shapes of one C2 training step, attention,
LayerNorm, Sinkhorn.  HIP events on the launch stream, interleaved rounds, median reported."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from timetuning_amd import hip_ops as ops  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts)


def main():
    dev = "cuda"
    M = 128 * 197
    Mt = 32 * 197
    print(f"{'op':34s} {'M':>6s} {'N':>5s} {'K':>5s}  {'us':>8s} {'TFLOP/s':>8s}")
    fwd = [("qkv", M, 1152, 384, 0), ("proj+res", M, 384, 384, 0), ("fc1+gelu", M, 1536, 384, 1), ("fc2+res", M, 384, 1536, 0),
           ("head0", 32 * 196, 1024, 384, 1), ("head2", 32 * 196, 1024, 1024, 1), ("scores", 32 * 196, 200, 256, 0)]
    tot = 0.0
    for name, m, n, k, act in fwd:
        x, w, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev) * 0.05, torch.randn(n, device=dev)
        res = torch.randn(m, n, device=dev) if "res" in name else None
        t = timeit(lambda: ops.linear_fwd(x, w, b, residual=res, act=act))
        print(f"fwd {name:30s} {m:6d} {n:5d} {k:5d}  {t * 1e6:8.1f} {2.0 * m * n * k / t / 1e12:8.1f}")
    for name, m, n, k in [("dgrad fc2", Mt, 384, 1536), ("dgrad fc1", Mt, 1536, 384), ("dgrad qkv", Mt, 1152, 384)]:
        dy, w = torch.randn(m, n, device=dev), torch.randn(n, k, device=dev) * 0.05
        t = timeit(lambda: ops.linear_bwd_data(dy, w))
        print(f"bwd {name:30s} {m:6d} {n:5d} {k:5d}  {t * 1e6:8.1f} {2.0 * m * n * k / t / 1e12:8.1f}")
    for name, m, n, k in [("wgrad fc2", Mt, 384, 1536), ("wgrad fc1", Mt, 1536, 384), ("wgrad qkv", Mt, 1152, 384), ("wgrad proj", Mt, 384, 384),
                          ("wgrad head2", 32 * 196, 1024, 1024)]:
        dy, x = torch.randn(m, n, device=dev), torch.randn(m, k, device=dev)
        t = timeit(lambda: ops.linear_bwd_weight(dy, x))
        print(f"bwd {name:30s} {m:6d} {n:5d} {k:5d}  {t * 1e6:8.1f} {2.0 * m * n * k / t / 1e12:8.1f}")
    qkv = torch.randn(128, 197, 1152, device=dev)
    t = timeit(lambda: ops.attention_fwd(qkv, 6))
    print(f"attention fwd F=128 N=197 H=6              {t * 1e6:8.1f} {128 * 6 * 4.0 * 197 * 197 * 64 / t / 1e12:8.1f}")
    x = torch.randn(128, 197, 384, device=dev)
    g, b = torch.ones(384, device=dev), torch.zeros(384, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b))
    print(f"layernorm fwd [25216,384]                  {t * 1e6:8.1f} {2 * x.numel() * 4 / t / 1e9:8.0f} GB/s")
    sc = torch.nn.functional.normalize(torch.randn(6272, 256, device=dev), dim=1) @ torch.nn.functional.normalize(torch.randn(200, 256, device=dev), dim=1).t()
    t = timeit(lambda: ops.sinkhorn(sc, 10))
    print(f"sinkhorn K=200 B=6272 10 it                {t * 1e6:8.1f} {10 / t:8.0f} it/s")


if __name__ == "__main__":
    main()
