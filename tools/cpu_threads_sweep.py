#!/usr/bin/env python3
"""Host-CPU baseline vs thread count: the oracle's training step (reference structure, 2 clips x 4 frames of ViT-S/16, K=200) at
torch.set_num_threads(t).  Shows why bench.py's cpu_baseline leg caps the threads at 32.  (Test infrastructure: uses oracle/.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import timet_oracle as O
from timetuning_amd import synth

om = O.build_oracle("dino-s16", 200, (1024, 1024, 512, 256), mode="dino")
x = torch.from_numpy(synth.make_clips(2, 4, 224, seed=1))
print(f"host cores: {os.cpu_count()}")
for t in (8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 1): break
    torch.set_num_threads(t)
    om.get_loss(x, faithful=True).backward()
    t0 = time.perf_counter()
    for _ in range(2):
        om.get_loss(x, faithful=True).backward()
    dt = (time.perf_counter() - t0) / 2
    print(f"threads {t:4d}  {dt:7.3f} s/step  {8 / dt:7.2f} clip-frames/s", flush=True)
