"""tests/test_hip_timet.py::test_two_streams_equal_one_stream, K-split on, with the hard labels of every step kept: is a gradient
difference between the one-stream and the two-stream run a label flip (a near-tie of the propagated maps within fp32 rounding) or a race?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import timetuning_amd  # noqa: F401
from timetuning_amd import engine, hip_ops, synth
from tools.graph_vs_eager import CONFIGS, make

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
arch, bs, fs, K, queue = CONFIGS[cfg]
hip_ops.set_gemm_precision("f16x3")
x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=12)).cuda()
for rep in range(3):
    runs = []
    for two in (False, True):
        engine.TWO_STREAMS = two
        m, o = make(cfg, 8)
        torch.manual_seed(5)
        rec = []
        for i in range(3):
            loss = m.get_loss(x)
            lab = m.last_aux["labels"].clone()
            m.train_update(o, loss, i + 1 if queue else 0)
            rec.append((loss.item(), lab, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
        runs.append(rec)
    for i in range(3):
        (l1, a1, g1), (l2, a2, g2) = runs[0][i], runs[1][i]
        worst = max((((g2[n] - g1[n]).double().norm() / g1[n].double().norm()).item(), n) for n in g1)
        print(f"rep {rep} step {i}: loss {l1:.6f} / {l2:.6f}; labels differing {int((a1 != a2).sum())} of {a1.numel()}; worst gradient rel L2 {worst[0]:.2e} ({worst[1]})", flush=True)
