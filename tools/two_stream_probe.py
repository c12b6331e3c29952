#!/usr/bin/env python3
"""Would the frozen blocks run faster as TWO half batches on two streams (the tail of one half's persistent GEMM filled by the other
half's next kernel)?  tt_vit_forward over 10 blocks of ViT-S/16 in the f16x3 mode: 128 frames on one stream against 2 x 64 frames on two
streams (and 4 x 32 on four), same box, interleaved."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from timetuning_amd import engine, hip_ops as ops

dev = torch.device("cuda", 0)
ops.set_gemm_precision("f16x3")
model = bench.build_model("dino-s16", 200, dev)
vit = model.feature_extractor.backbone
F, N, D, NB = 128, 197, 384, 10
tok = torch.randn(F, N, D, device=dev)
params = engine.vit_params(vit, 0, NB)

def run(parts, streams):
    h = F // parts
    cur = torch.cuda.current_stream()
    for i in range(parts):
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            ops.vit_forward(params, NB, work[i * h:(i + 1) * h])
    for i in range(parts):
        cur.wait_stream(streams[i])

streams = [torch.cuda.Stream() for _ in range(4)]
res = {1: [], 2: [], 4: []}
for rd in range(8):
    for parts in (1, 2, 4):
        work = tok.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if parts == 1:
            ops.vit_forward(params, NB, work)
        else:
            run(parts, streams)
        e1.record(); torch.cuda.synchronize()
        if rd >= 2: res[parts].append(e0.elapsed_time(e1))
for parts, v in res.items():
    print(f"{parts} stream(s) x {F // parts} frames: {statistics.median(v):.3f} ms for {NB} frozen blocks (min {min(v):.3f})")

# ---- the trainable blocks (10, 11): the frames that keep nothing (96) and the kept target frames (32) are independent chains
print("trainable blocks 10-11, forward: rest chain (96 frames) and kept chain (32 frames)")
blocks = [vit.blocks[10], vit.blocks[11]]
def chains(concurrent):
    x = tok.clone()
    lo, hi = x[:96], x[96:]
    cur = torch.cuda.current_stream()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    saves = [{}, {}]
    if concurrent:
        a, b = streams[0], streams[1]
        a.wait_stream(cur); b.wait_stream(cur)
        with torch.cuda.stream(a):
            for blk in blocks: lo = engine.block_forward(lo, blk, vit.num_heads, None, None)
        with torch.cuda.stream(b):
            for i, blk in enumerate(blocks): hi = engine.block_forward(hi, blk, vit.num_heads, saves[i], None)
        cur.wait_stream(a); cur.wait_stream(b)
    else:
        for i, blk in enumerate(blocks):
            lo = engine.block_forward(lo, blk, vit.num_heads, None, None)
            hi = engine.block_forward(hi, blk, vit.num_heads, saves[i], None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
r = {False: [], True: []}
for rd in range(8):
    for c in (False, True):
        t = chains(c)
        if rd >= 2: r[c].append(t)
print(f"one stream: {statistics.median(r[False]):.3f} ms   two streams: {statistics.median(r[True]):.3f} ms")
