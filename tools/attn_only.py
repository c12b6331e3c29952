import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from timetuning_amd import hip_ops as ops
qkv = torch.randn(128, 197, 1152, device="cuda")
for _ in range(5): ops.attention_fwd(qkv, 6)
torch.cuda.synchronize()
