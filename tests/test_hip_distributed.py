"""The N>1 path on real kernels with ONE GPU: two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one
device), each runs the fused training step on its own clips.  The data-parallel result must equal a single process on
the concatenated batch: the global Sinkhorn couples all columns, the loss is the mean over all clips, and the
all-reduced gradient is the gradient of that mean."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, HEAD, BS, FS = 20, (128, 128, 64, 32), 2, 2
WATCH = ("prototypes", "feature_extractor.head.6.weight", "feature_extractor.backbone.blocks.10.attn.qkv.weight",
         "feature_extractor.backbone.blocks.11.norm2.bias")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    from timetuning_amd import synth
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.time_tuning import TimeT

    fe = FeatureExtractor("dino-s16", "", list(HEAD), unfreeze_layers=["blocks.11", "blocks.10"], vit_cfg=synth.ARCHS["tiny-s16"],
                          init="stress", return_attention=False)
    return TimeT(fe, K, prototype_init=torch.from_numpy(synth.make_prototypes(K, fe.feature_dim))).cuda()


def _worker(rank, W, port, ret):
    import sys

    sys.path.insert(0, REPO)
    import torch.distributed as dist

    from timetuning_amd import synth
    from timetuning_amd.models import DistributedDataParallelModel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    model = DistributedDataParallelModel(_model(), 0)
    x = torch.from_numpy(synth.make_clips(BS, FS, 224, seed=11 + rank)).cuda()
    loss = model(x, None, True, False)
    loss.backward()
    params = dict(model.get_non_ddp_model().named_parameters())
    ret[rank] = dict(loss=float(loss.item()), grads={n: params[n].grad.cpu().numpy() for n in WATCH},
                     q=model.last_aux["q"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_equal_single_process_on_concatenated_batch():
    import torch.multiprocessing as mp

    from timetuning_amd import synth

    W = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, W, port, ret)) for r in range(W)]
    [p.start() for p in procs]
    [p.join(500) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]

    model = _model()
    x_all = torch.from_numpy(np.concatenate([synth.make_clips(BS, FS, 224, seed=11 + r) for r in range(W)], axis=0)).cuda()
    loss = model.get_loss(x_all)
    loss.backward()
    params = dict(model.named_parameters())
    assert abs(loss.item() - 0.5 * (ret[0]["loss"] + ret[1]["loss"])) < 1e-5
    q_all = model.last_aux["q"].cpu().numpy()
    for r in range(W):
        assert np.abs(q_all[r * BS:(r + 1) * BS] - ret[r]["q"]).max() < 1e-6 * max(1.0, np.abs(q_all).max())
    for n in WATCH:
        ref = params[n].grad.cpu().numpy()
        for r in range(W):
            err = np.abs(ret[r]["grads"][n] - ref).max() / np.abs(ref).max()
            assert err < 1e-4, (n, r, err)
