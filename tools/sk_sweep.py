import os, sys, statistics, torch
sys.path.insert(0, "/root/repo")
from timetuning_amd import hip_ops as ops
B = int(sys.argv[1]); K = 200
torch.manual_seed(0)
sc = torch.nn.functional.normalize(torch.randn(B, 256, device="cuda"), dim=1) @ torch.nn.functional.normalize(torch.randn(K, 256, device="cuda"), dim=1).t()
for _ in range(3): ops.sinkhorn(sc, 10, rows_out=6272)
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.sinkhorn(sc, 10, rows_out=6272); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"B={B} WGS={os.environ.get('TT_SK_WGS','default')}: {statistics.median(ts):.1f} us")
