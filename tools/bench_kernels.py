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
    # training-protocol propagation (C2: 32 clips x 4 frames, K=200, 7 context frames, radius 6)
    xn = torch.nn.functional.normalize(torch.randn(4, 32, 196, 384, device=dev), dim=-1)
    q0 = torch.softmax(torch.randn(32, 196, 200, device=dev), -1)
    t = timeit(lambda: ops.label_propagate(xn, q0))
    print(f"label_propagate C2 (32x4 fr, K=200)        {t * 1e6:8.1f} {32 * 3 / t:8.0f} frames/s")
    # --use_mask: foreground masks of 64 frames from the last block's qkv
    qkv64 = torch.randn(64, 197, 1152, device=dev)
    t = timeit(lambda: ops.foreground_mask(qkv64, 6, 14))
    print(f"foreground_mask 64 frames g=14             {t * 1e6:8.1f} {64 / t:8.0f} frames/s")
    # evaluator clustering (N2): 64 frames of 196 x 50 PCA features -> 224 x 224, then one k-means assignment scan (k = 21)
    f50 = torch.randn(64, 196, 50, device=dev)
    t = timeit(lambda: ops.upsample_bilinear_tokens(f50, 224))
    up = ops.upsample_bilinear_tokens(f50, 224).view(-1, 50)
    print(f"upsample tokens 64 x 14^2 -> 224^2 x 50     {t * 1e6:8.1f} {up.numel() * 4 / t / 1e9:8.0f} GB/s written")
    cent = up[:21].clone()
    t = timeit(lambda: ops.kmeans_assign(up, cent))
    print(f"kmeans_assign {up.shape[0]} pts d=50 k=21        {t * 1e6:8.1f} {up.numel() * 4 / t / 1e9:8.0f} GB/s read")
    lab = ops.kmeans_assign(up, cent)
    t = timeit(lambda: ops.kmeans_accumulate(up, lab, 21))
    print(f"kmeans_accumulate (deterministic)          {t * 1e6:8.1f} {up.numel() * 4 / t / 1e9:8.0f} GB/s read")
    # input pipeline (N3): 4-frame 480x854 uint8 clips -> jitter/blur at full size -> Resize(224) -> RandomResizedCrop -> tensor
    import random
    from timetuning_amd import video_transformations as VT
    frames = torch.randint(0, 256, (4, 480, 854, 3), dtype=torch.uint8, device=dev)
    dt, vt_ = VT.training_transforms(224)
    random.seed(0); torch.manual_seed(0)
    t = timeit(lambda: vt_(dt(frames)), reps=30)
    print(f"input pipeline, one 4-frame 480x854 clip   {t * 1e6:8.1f} {4 / t:8.0f} frames/s (random branches, mean of 30 draws)")
    t = timeit(lambda: vt_(frames), reps=30)
    print(f"  video_transform only (resize+crop+tensor)  {t * 1e6:8.1f} {4 / t:8.0f} frames/s")
    # evaluation protocol (N4): one 25-frame clip, 4 context frames, radius 12, 8 classes, + upsample/argmax to 224
    for g_ in (14, 28):
        xn = torch.nn.functional.normalize(torch.randn(25, 1, g_ * g_, 384, device=dev), dim=-1)
        seed = torch.nn.functional.one_hot(torch.randint(0, 8, (1, g_ * g_), device=dev), 8).float()
        t = timeit(lambda: ops.label_propagate_maps(xn, seed, 4, 12, 5))
        maps = ops.label_propagate_maps(xn, seed, 4, 12, 5).view(24, g_ * g_, 8)
        t2 = timeit(lambda: ops.upsample_argmax(maps, 224))
        print(f"eval propagate 25 fr g={g_:2d} r=12 + upsample   {t * 1e6:8.1f} {24 / (t + t2):8.0f} frames/s (upsample+argmax {t2 * 1e6:.1f} us)")


if __name__ == "__main__":
    main()
