"""The compute stream's exposed waits on each collective of the step on a ONE-rank RCCL communicator, with the streams as shipped
(bench.py's instrumented step runs on one stream).  TT_SINGLE_STREAM=1 for the one-stream figures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import timetuning_amd  # noqa
import torch, torch.distributed as dist
import bench
from timetuning_amd import engine, hip_ops as ops, synth
from timetuning_amd.my_utils import cosine_scheduler
from timetuning_amd.time_tuning import SwavOptimizer

os.environ["TT_EXCHANGE_SINGLE_RANK"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0, device_id=dev)
ops.set_gemm_precision("f16x3")
model = bench.build_model("dino-s16", 200, dev)
opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 300), 300, 1)
x = torch.from_numpy(synth.make_clips(32, 4, 224, seed=1)).to(dev)
for _ in range(5): bench.train_step(model, opt, x, False)
torch.cuda.synchronize()
engine.RCCL_PROFILE = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bench.train_step(model, opt, x, False); e1.record(); torch.cuda.synchronize()
waits, engine.RCCL_PROFILE = engine.RCCL_PROFILE, None
print(f"step {e0.elapsed_time(e1):.3f} ms")
for k, n, a, b in waits: print(f"  {k:32s} {n:10d} B  exposed wait {a.elapsed_time(b) * 1e3:8.1f} us")
dist.destroy_process_group()
