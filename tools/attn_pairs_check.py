"""Round 4: the pair attention kernel (attention_pairs.hip) against fp64, next to the f32 kernel on the same qkv, with interleaved timings."""
import sys
import torch

sys.path.insert(0, ".")
from timetuning_amd import hip_ops as ops, synth  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()


for Fr, N, H in [(3, 197, 6), (2, 50, 2), (1, 256, 12), (2, 225, 3), (5, 17, 1), (128, 197, 6), (128, 197, 12),
                  (2, 785, 6), (1, 300, 2), (128, 785, 6), (32, 785, 12)]:   # > 256 tokens: the KV-tiled kernel (785 = C5)
    D = 64 * H
    qkv = torch.from_numpy(synth.normal(f"attp.{Fr}.{N}.{H}", (Fr, N, 3 * D), 1.0)).cuda() * (1.5 if Fr < 30 else 1.0)
    qkvp = ops.split_pairs(qkv.view(Fr * N, 3 * D)).view(Fr, N, 6 * D)
    q, k, v = qkv.double().view(Fr, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) * 64 ** -0.5
    ref = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(Fr, N, D)
    ref_lse = torch.logsumexp(sc, dim=-1)
    op, of, lse = ops.attention_fwd_pairs(qkvp, H, out_pairs=True, out_f32=True, save_lse=True)
    o32, lse32, _ = ops.attention_fwd(qkv, H, save_lse=True)
    same = all(torch.equal(ops.attention_fwd_pairs(qkvp, H, out_f32=True)[1], of) for _ in range(3))
    print(f"F={Fr} N={N} H={H}: pairs kernel fp32 out {rel(of, ref)}  pairs out {rel(ops.join_pairs(op), ref)}  lse {rel(lse, ref_lse)} | f32 kernel {rel(o32, ref)} lse {rel(lse32, ref_lse)}"
          f"  repeat={'ok' if same else 'DIFFERS'}", flush=True)
    if Fr >= 30:
        def t(fn, n=20):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        def tiled():
            ops.set_tuning_knob("TT_ATTN_PAIRS_FLASH", 1)
            ops.attention_fwd_pairs(qkvp, H)
            ops.set_tuning_knob("TT_ATTN_PAIRS_FLASH", 0)
        fns = {"pairs (pairs out)": lambda: ops.attention_fwd_pairs(qkvp, H), "pairs (all outputs)": lambda: ops.attention_fwd_pairs(qkvp, H, True, True, True),
               "pairs, KV-tiled kernel forced": tiled, "f32": lambda: ops.attention_fwd(qkv, H)}
        best = {kk: 1e9 for kk in fns}
        for _ in range(3):
            for kk, fn in fns.items():
                best[kk] = min(best[kk], t(fn))
        fl = 4.0 * Fr * H * N * N * 64
        print("        us (TFLOP/s-equivalent): " + "  ".join(f"{kk} {vv:.1f} ({fl / vv * 1e-6:.0f})" for kk, vv in best.items()), flush=True)
