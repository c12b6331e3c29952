"""Replays of a captured step WITHOUT an update in between must all be the same step (round 6 debugging aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timetuning_amd import hip_ops, synth
from timetuning_amd.models import FeatureExtractor
from timetuning_amd.time_tuning import TimeT

hip_ops.set_gemm_precision("f16x3")
bs, fs, K = 32, 4, 200
x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=3)).cuda()
for init in ("stress", "dino"):
    for with_labels in (False, True):
        for backward in (True, False):
            fe = FeatureExtractor("dino-s16", "", [1024, 1024, 512, 256], unfreeze_layers=["blocks.11", "blocks.10"], init=init, return_attention=False)
            m = TimeT(fe, K, prototype_init=torch.from_numpy(synth.make_prototypes(K, 256))).cuda()
            m.enable_step_graph()
            labels = torch.randint(0, K, (bs, 196)) if with_labels else None
            out = []
            for call in range(4):
                m.zero_grad(set_to_none=True)
                loss = m.get_loss(x, target_labels=labels)
                ts = m.last_aux["target_scores"].double()
                out.append((round(loss.item(), 6), round(ts.abs().sum().item(), 3), round(m.last_aux["q"].double().abs().sum().item(), 6)))
                if backward:
                    loss.backward()
                    out[-1] += (round(m.prototypes.grad.double().abs().sum().item(), 6),)
            print(f"init={init} labels={with_labels} backward={backward}: " + " | ".join(map(str, out)), flush=True)
