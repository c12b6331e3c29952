import sys, torch
sys.path.insert(0, ".")
from timetuning_amd import hip_ops as ops
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K, name in ((25216, 384, 384, "proj"), (25216, 384, 1536, "fc2"), (25216, 1152, 384, "qkv"), (25216, 768, 768, "B proj"), (25216, 768, 3072, "B fc2")):
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
    xp, wp = ops.split_pairs(x), ops.split_pairs(w)
    y = torch.randn(M, N, device="cuda")
    best = {0: 1e9, 1: 1e9}
    for _ in range(4):
        for nh in (0, 1):
            ops.set_tuning_knob("TT_P8_NO_HALF", nh)
            best[nh] = min(best[nh], t(lambda: ops.linear_fwd_pairs(xp, wp, b, residual=y, out=y)))
    ops.set_tuning_knob("TT_P8_NO_HALF", 0)
    print(f"{name:7s} M={M} N={N} K={K}: half tiles {best[0]:.1f} us | whole tiles only {best[1]:.1f} us", flush=True)
