"""Child process of tests/test_hip_distributed.py::test_one_rank_exchange_path_equals_no_exchange: the exchange path (one-rank RCCL
communicator, TT_EXCHANGE_SINGLE_RANK=1: the very calls of an N-GPU rank) against the plain one-process step, and gradient accumulation
across steps on the double-buffered gradient arena.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import timetuning_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from timetuning_amd import hip_ops, synth  # noqa: E402
from timetuning_amd.models import FeatureExtractor  # noqa: E402
from timetuning_amd.my_utils import cosine_scheduler  # noqa: E402
from timetuning_amd.time_tuning import SwavOptimizer, TimeT  # noqa: E402


def make():
    fe = FeatureExtractor("dino-s16", "", [1024, 1024, 512, 256], unfreeze_layers=["blocks.11", "blocks.10"], init="dino", return_attention=False)
    m = TimeT(fe, 200, prototype_init=torch.from_numpy(synth.make_prototypes(200, 256))).cuda()
    o = SwavOptimizer(m, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 12), 12, 1)
    return m, o


def main():
    torch.cuda.set_device(0)
    hip_ops.set_gemm_precision("f16x3")
    bs, fs = 8, 4
    clips = [torch.from_numpy(synth.make_clips(bs, fs, 224, seed=30 + i)).cuda() for i in range(4)]
    out = {}
    # (a) plain process: four steps; and the gradients of clips 0 / 1 from the initial weights (the accumulation reference)
    m, o = make()
    g_ref = []
    for i in range(2):
        m.zero_grad(set_to_none=True)
        m.get_loss(clips[i]).backward()
        g_ref.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    m, o = make()
    plain = []
    for x in clips:
        loss = m.get_loss(x)
        m.train_update(o, loss, 0)
        plain.append(loss.item())
    p_plain = {n: p.detach().clone() for n, p in m.named_parameters()}
    # (b) the exchange path on a one-rank communicator
    os.environ["TT_EXCHANGE_SINGLE_RANK"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    m, o = make()
    ex = []
    for x in clips:
        loss = m.get_loss(x)
        m.train_update(o, loss, 0)
        ex.append(loss.item())
    arena = m._grad_arena
    out["losses_equal"] = plain == ex
    out["params_equal"] = all(torch.equal(p_plain[n], p.detach()) for n, p in m.named_parameters())
    out["two_buffers"] = len(arena.flats) == 2
    lo = [f.data_ptr() for f in arena.flats]
    hi = [f.data_ptr() + 4 * f.numel() for f in arena.flats]
    out["grads_in_current_buffer"] = all(lo[arena.cur] <= p.grad.data_ptr() < hi[arena.cur] for p in m.parameters() if p.requires_grad)
    # (c) accumulation across steps WITHOUT zero_grad on the arena: g0 + g1 (three backward calls: the third overwrites the buffer the first used)
    m, o = make()
    m.get_loss(clips[0]).backward()      # first step: hand-flattened, builds the arena
    m.zero_grad(set_to_none=True)
    m.get_loss(clips[0]).backward()      # arena buffer 1
    m.get_loss(clips[1]).backward()      # arena buffer 0, accumulated into .grad (a view of buffer 1)
    acc = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.get_loss(clips[1]).backward()      # buffer 1 again: the accumulated .grad aliases it and must have been given its own storage
    acc2 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    err = max(float((acc[n] - (g_ref[0][n] + g_ref[1][n])).abs().max() / (g_ref[0][n] + g_ref[1][n]).abs().max().clamp_min(1e-30)) for n in acc)
    err2 = max(float((acc2[n] - (g_ref[0][n] + 2 * g_ref[1][n])).abs().max() / (g_ref[0][n] + 2 * g_ref[1][n]).abs().max().clamp_min(1e-30)) for n in acc2)
    out["accumulate_err"] = err
    out["accumulate3_err"] = err2
    torch.cuda.synchronize()
    print("RESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
