import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench
from timetuning_amd import hip_ops as ops, synth
dev = torch.device("cuda", 0)
x = torch.from_numpy(synth.make_clips(32, 4, 224, seed=1)).to(dev)
res = {}
for mode in ("f32", "f16x3"):
    ops.set_gemm_precision(mode)
    torch.manual_seed(0)
    model = bench.build_model("dino-s16", 200, dev)
    model.zero_grad(set_to_none=True)
    loss = model(x, None, True, False); loss.backward(); torch.cuda.synchronize()
    res[mode] = (loss.item(), {n: p.grad.double().clone() for n, p in model.named_parameters() if p.grad is not None})
print("loss", res["f32"][0], res["f16x3"][0])
rows = []
for n, g32 in res["f32"][1].items():
    g16 = res["f16x3"][1][n]
    rows.append((((g16 - g32).norm() / g32.norm()).item(), ((g16 - g32).abs().max() / g32.abs().max()).item(), g32.abs().max().item(), g32.abs().median().item(), n))
rows.sort(reverse=True)
for r in rows[:12]: print("rel-L2 %.2e  max-norm %.2e  |g|max %.2e median %.2e  %s" % r)
# magnitudes of the dy tensors the pair split sees: hook split_pairs_dual
mags = []
real = ops.split_pairs_dual
def hook(x_, *a, **k):
    if x_.shape[0] > 1000: mags.append((tuple(x_.shape), x_.abs().max().item(), x_.abs().median().item(), (x_.abs() < 6.1e-5).float().mean().item()))
    return real(x_, *a, **k)
ops.split_pairs_dual = hook
model.zero_grad(set_to_none=True); loss = model(x, None, True, False); loss.backward(); torch.cuda.synchronize()
for m in mags: print("dy %s max %.2e median %.2e  fraction below fp16's smallest normal %.2f" % m)
