import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timetuning_amd import hip_ops, synth
from timetuning_amd.models import FeatureExtractor
from timetuning_amd.time_tuning import TimeT

hip_ops.set_gemm_precision("f16x3")
bs, fs, K = 32, 4, 200
x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=3)).cuda()
def mk():
    fe = FeatureExtractor("dino-s16", "", [1024, 1024, 512, 256], unfreeze_layers=["blocks.11", "blocks.10"], init="stress", return_attention=False)
    return TimeT(fe, K, prototype_init=torch.from_numpy(synth.make_prototypes(K, 256))).cuda()
m0 = mk()
m0.get_loss(x)
labels = m0.last_aux["labels"].cpu().clone()
ref_ts = m0.last_aux["target_scores"].clone()
del m0
for variant in ("cpu_labels", "grads_cpu", "grads_clone", "aux_cpu"):
    m = mk()
    m.enable_step_graph()
    prev = None
    for call in range(4):
        m.zero_grad(set_to_none=True)
        loss = m.get_loss(x, target_labels=labels if variant != "gpu_labels" else labels.cuda())
        ts = m.last_aux["target_scores"].clone()
        q = m.last_aux["q"].clone()
        d = (ts - ref_ts).abs()
        rows = (d.view(-1, K).max(dim=1).values > 0).nonzero().view(-1)
        print(variant, call, "loss", round(loss.item(), 6), "ts vs eager ref: max abs", d.max().item(), "rows differing", rows.numel(),
              (rows[:5].tolist(), rows[-5:].tolist()) if rows.numel() else "", flush=True)
        if variant == "aux_cpu":
            _ = m.last_aux["q"].cpu(); _ = m.last_aux["target_scores"].cpu(); _ = m.last_aux["labels"].cpu()
        loss.backward()
        if variant == "grads_cpu":
            _ = [p.grad.cpu() for p in m.parameters() if p.grad is not None]
        if variant == "grads_clone":
            _ = [p.grad.clone() for p in m.parameters() if p.grad is not None]
