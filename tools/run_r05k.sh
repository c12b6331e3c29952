#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05k
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "sinkhorn" 2>&1 | grep -v "amdgpu.ids" | tail -15 > "$O/tests_sk.log"
cat "$O/tests_sk.log"
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee "$O/sk_rate.txt"
import torch, sys
sys.path.insert(0, ".")
from timetuning_amd import hip_ops as ops, synth
import torch.nn.functional as F
def rate(B, K, iters=10, reps=50):
    x = F.normalize(torch.from_numpy(synth.normal("bench.sk.x", (B, 256))), dim=1); p = torch.from_numpy(synth.make_prototypes(K, 256))
    s = (x @ p.t()).cuda()
    for _ in range(5): ops.sinkhorn(s, iters)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.sinkhorn(s, iters)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
import os
for B in (6272, 8320, 50176, 66560):
    for rows in (0,):
        res = {}
        for mode in (1, 0):
            ops.set_tuning_knob("TT_SK_PERSIST", mode)
            res[mode] = rate(B, 200)
        print(f"B={B} K=200 10 iterations: one launch {res[1]:7.1f} us ({1e7/res[1]:8.0f} iters/s) | launch per iteration {res[0]:7.1f} us ({1e7/res[0]:8.0f} iters/s)", flush=True)
ops.set_tuning_knob("TT_SK_PERSIST", 1)
PY
for r in 56 84 112 140 168; do TT_SKP_ROWS=$r python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a "$O/sk_rate.txt"
import torch, sys, os
sys.path.insert(0, ".")
from timetuning_amd import hip_ops as ops, synth
import torch.nn.functional as F
x = F.normalize(torch.from_numpy(synth.normal("bench.sk.x", (6272, 256))), dim=1); p = torch.from_numpy(synth.make_prototypes(200, 256))
s = (x @ p.t()).cuda()
for _ in range(5): ops.sinkhorn(s, 10)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.sinkhorn(s, 10)
e1.record(); torch.cuda.synchronize()
print(f"TT_SKP_ROWS={os.environ['TT_SKP_ROWS']}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per 10-iteration solve at B=6272")
PY
done
