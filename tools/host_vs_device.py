"""How long does the HOST need to issue a C2 step's launches (the loop without a final synchronisation) against the device's step time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import timetuning_amd  # noqa
import torch
import bench
from timetuning_amd import hip_ops as ops, synth
from timetuning_amd.my_utils import cosine_scheduler
from timetuning_amd.time_tuning import SwavOptimizer

ops.set_gemm_precision("f16x3")
dev = torch.device("cuda", 0)
if os.environ.get("TT_EXCHANGE_SINGLE_RANK") == "1":   # the exchange path on a one-rank RCCL communicator (bench.py's probe child)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0, device_id=dev)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = bench.build_model("dino-s16", 200, dev)
opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 300), 300, 1)
x = torch.from_numpy(synth.make_clips(bs, 4, 224, seed=1)).to(dev)
for _ in range(5): bench.train_step(model, opt, x, False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): bench.train_step(model, opt, x, False)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"bs={bs}: host issue {t_host / N * 1e3:.3f} ms per step, device-complete {t_all / N * 1e3:.3f} ms per step")
