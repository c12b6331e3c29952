import os, sys, statistics, torch
sys.path.insert(0, "/root/repo")
from timetuning_amd import hip_ops as ops
M, N, K = 6272, 200, 256
x = torch.nn.functional.normalize(torch.randn(M, K, device="cuda"), dim=1); w = torch.nn.functional.normalize(torch.randn(N, K, device="cuda"), dim=1)
ds = torch.randn(M, N, device="cuda")
def t(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); 
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 200)
    return statistics.median(ts)
print("TT_FORCE_TILE", os.environ.get("TT_FORCE_TILE"), "fwd scores %.1f us" % t(lambda: ops.linear_fwd(x, w)), "dgrad %.1f us" % t(lambda: ops.linear_bwd_data(ds, w)),
      "wgrad %.1f us" % t(lambda: ops.linear_bwd_weight(ds, x, need_bias=False)))
