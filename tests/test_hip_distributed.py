"""The N>1 path on real kernels.  Two ranks run the fused training step on their own clips; the data-parallel result must
equal a single process on the concatenated batch: the global Sinkhorn couples all columns, the loss is the mean over all clips,
and the all-reduced gradient is the gradient of that mean.

* ``gloo``: both ranks share cuda:0 (RCCL refuses two ranks on one device) - runs on the 1-GPU box.
* ``nccl`` (= RCCL): one rank per GPU, needs >= 2 visible GPUs (skipped otherwise).
* the RCCL calls themselves (async all_gather_into_tensor + bucketed async all_reduce on side streams) are also executed on
  the 1-GPU box through a ONE-rank ``nccl`` group with ``TT_EXCHANGE_SINGLE_RANK=1`` (engine.exchange_group).
* ``bench.py --gpus 2`` is run through its own launcher (two ranks on cuda:0 over gloo)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, HEAD, BS, FS = 20, (128, 128, 64, 32), 2, 2
SK_ITERS = 10   # get_loss default (time_tuning.py:327)
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


def _worker(rank, W, port, ret, backend="gloo"):
    import sys

    sys.path.insert(0, REPO)
    import torch.distributed as dist

    from timetuning_amd import synth
    from timetuning_amd.models import DistributedDataParallelModel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=W)
    model = DistributedDataParallelModel(_model(), dev)
    x = torch.from_numpy(synth.make_clips(BS, FS, 224, seed=11 + rank)).cuda()
    from timetuning_amd import engine

    # step 1 flattens its buckets by hand and records their layout; step 2 (same data, gradients cleared) must give the same numbers with
    # the backward kernels writing the persistent flat buckets directly (engine.GradArena) - and no torch.cat
    inner = model.get_non_ddp_model()
    params = dict(inner.named_parameters())
    loss = model(x, None, True, False)
    loss.backward()
    first = {n: params[n].grad.clone() for n in WATCH}
    inner.zero_grad(set_to_none=True)
    cats = {"n": 0}
    real_cat = torch.cat

    def counting_cat(*a, **k):
        cats["n"] += 1
        return real_cat(*a, **k)

    engine.RCCL_PROFILE = []   # every wait of the compute stream on a collective: (kind, payload bytes, events)
    torch.cat = counting_cat
    try:
        loss = model(x, None, True, False)
        loss.backward()
    finally:
        torch.cat = real_cat
    torch.cuda.synchronize()
    waits, engine.RCCL_PROFILE = engine.RCCL_PROFILE, None
    arena = inner._grad_arena
    lo, hi = arena.flat.data_ptr(), arena.flat.data_ptr() + arena.flat.numel() * 4
    # step 3, forward only: the reference's own exchange pattern for the assignment (--sinkhorn_exchange allreduce, my_utils.py:250-272) -
    # the HIP kernels of one rank's share (tt_sinkhorn_local_*) with the K row sums all-reduced once per iteration
    q_gather = model.last_aux["q"].clone()
    engine.SINKHORN_EXCHANGE = "allreduce"
    engine.RCCL_PROFILE = []
    try:
        with torch.no_grad():
            loss_r = model(x, None, True, False)
    finally:
        engine.SINKHORN_EXCHANGE = "allgather"
    torch.cuda.synchronize()
    waits_r, engine.RCCL_PROFILE = engine.RCCL_PROFILE, None
    arena_stats = dict(arena_floats=arena.flat.numel(), bucket_sizes=[e - s_ for s_, e, _ in arena.buckets],
                       in_arena=all(lo <= p.grad.data_ptr() < hi for p in inner.parameters() if p.requires_grad),
                       same_as_first=all(torch.equal(params[n].grad, first[n]) for n in WATCH))
    # step 4 ...: the exchange decides for itself (engine.autotune_exchange, what time_tuning() / bench.py run on the first multi-rank batch):
    # the three configurations are timed on the real step, every rank must arrive at the same choice, and a step under the chosen
    # configuration (whichever it is - one bucket, the all-reduce Sinkhorn) still gives step 1's loss and gradients
    def _probe():
        inner.zero_grad(set_to_none=True)
        model(x, None, True, False).backward()

    choice = engine.autotune_exchange(_probe, torch.device("cuda", dev), reps=2)
    inner.zero_grad(set_to_none=True)
    loss_c = model(x, None, True, False)
    loss_c.backward()
    torch.cuda.synchronize()
    auto = dict(choice=(choice["sinkhorn_exchange"], choice["grad_buckets"]), keys=sorted(choice["ms_per_step"]),
                set=(engine.SINKHORN_EXCHANGE, engine.GRAD_BUCKETS), loss=float(loss_c.item()),
                grad_err=max(float((params[n].grad - first[n]).abs().max() / first[n].abs().max()) for n in WATCH))
    engine.SINKHORN_EXCHANGE, engine.GRAD_BUCKETS, engine.EXCHANGE_CHOICE = "allgather", 0, None
    ret[rank] = dict(auto=auto, loss=float(loss.item()), loss_allreduce=float(loss_r.item()),
                     q_allreduce_err=float((model.last_aux["q"] - q_gather).abs().max()),
                     waits_allreduce=[(k, n) for k, n, _, _ in waits_r if "sinkhorn" in k or "gather" in k], grads={n: params[n].grad.cpu().numpy() for n in WATCH},
                     q=model.last_aux["q"].cpu().numpy(), waits=[(k, n, e0.elapsed_time(e1)) for k, n, e0, e1 in waits],
                     trainable=sum(p.numel() for p in inner.parameters() if p.requires_grad),
                     cats=cats["n"], **arena_stats)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("backend,W", [("gloo", 2), ("gloo", 4), ("gloo", 8), ("nccl", 2)])
def test_two_ranks_equal_single_process_on_concatenated_batch(backend, W):
    """W ranks (2, and a 4- / 8-rank soak on the shared device) against ONE process on the concatenated batch: loss, assignment, gradients;
    the exchange each rank issued (bucket order, bytes); and the persistent flat buckets of the second step."""
    import torch.multiprocessing as mp

    from timetuning_amd import synth

    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL two-rank run needs two GPUs (one rank per device)")
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, W, port, ret, backend)) for r in range(W)]
    [p.start() for p in procs]
    [p.join(500) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    # the exchange each rank issued: ONE all-gather of the score rows of all ranks, FOUR gradient buckets that carry every trainable
    # float exactly once, and a (non-negative) exposed wait of the compute stream measured for each
    for r in range(W):
        waits = ret[r]["waits"]
        assert [k for k, _, _ in waits] == ["all_gather(scores)"] + [f"all_reduce(grad bucket {i})" for i in range(4)], waits
        assert waits[0][1] == W * BS * 196 * K * 4
        assert sum(n for _, n, _ in waits[1:]) == 4 * ret[r]["trainable"]
        assert all(ms >= 0 for _, _, ms in waits)
        # step 2 ran on the persistent arena: every trainable float once, bucket by bucket in the order of the waits, the gradients the
        # optimizer sees are views of it, nothing was concatenated, and the numbers equal step 1's bit for bit
        assert ret[r]["arena_floats"] == ret[r]["trainable"] and [4 * b for b in ret[r]["bucket_sizes"]] == [n for _, n, _ in waits[1:]]
        assert ret[r]["in_arena"] and ret[r]["cats"] == 0 and ret[r]["same_as_first"], (ret[r]["in_arena"], ret[r]["cats"], ret[r]["same_as_first"])
        # the all-reduce variant of the assignment: no all-gather, one K-float all-reduce per Sinkhorn iteration, the same assignment and loss
        assert ret[r]["waits_allreduce"] == [("all_reduce(sinkhorn row sums)", K * 4)] * SK_ITERS, ret[r]["waits_allreduce"]
        assert ret[r]["q_allreduce_err"] < 2e-6 and abs(ret[r]["loss_allreduce"] - ret[r]["loss"]) < 1e-5, (ret[r]["q_allreduce_err"], ret[r]["loss_allreduce"], ret[r]["loss"])
        # the self-deciding exchange: one decision for the whole job, consistent state, unchanged numbers under it
        au = ret[r]["auto"]
        assert au["choice"] == ret[0]["auto"]["choice"] and au["choice"][0] in ("allgather", "allreduce") and au["choice"][1] in (1, 4), au
        assert au["set"] == (au["choice"][0], 0 if au["choice"][1] == 4 else 1) and len(au["keys"]) == 3 and "allgather/4" in au["keys"] and "allreduce/4" in au["keys"], au
        assert abs(au["loss"] - ret[r]["loss"]) < 1e-5 and au["grad_err"] < 1e-4, au

    model = _model()
    x_all = torch.from_numpy(np.concatenate([synth.make_clips(BS, FS, 224, seed=11 + r) for r in range(W)], axis=0)).cuda()
    loss = model.get_loss(x_all)
    loss.backward()
    params = dict(model.named_parameters())
    assert abs(loss.item() - sum(ret[r]["loss"] for r in range(W)) / W) < 1e-5
    q_all = model.last_aux["q"].cpu().numpy()
    for r in range(W):
        assert np.abs(q_all[r * BS:(r + 1) * BS] - ret[r]["q"]).max() < 1e-6 * max(1.0, np.abs(q_all).max())
    for n in WATCH:
        ref = params[n].grad.cpu().numpy()
        for r in range(W):
            err = np.abs(ret[r]["grads"][n] - ref).max() / np.abs(ref).max()
            assert err < 1e-4, (n, r, err)


def _single_rank_nccl_worker(port, ret):
    """ONE rank, backend nccl: with TT_EXCHANGE_SINGLE_RANK=1 the step issues the same RCCL calls as on N GPUs."""
    import sys

    sys.path.insert(0, REPO)
    import torch.distributed as dist

    from timetuning_amd import engine, synth
    from timetuning_amd.models import DistributedDataParallelModel
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TT_EXCHANGE_SINGLE_RANK="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert engine.exchange_group() is not None and dist.get_backend() == "nccl"
    calls = {"all_gather": 0, "all_reduce": 0}
    ag, ar = dist.all_gather_into_tensor, dist.all_reduce

    def count_ag(*a, **k):
        calls["all_gather"] += 1
        return ag(*a, **k)

    def count_ar(*a, **k):
        calls["all_reduce"] += 1
        return ar(*a, **k)

    dist.all_gather_into_tensor, dist.all_reduce = count_ag, count_ar
    model = DistributedDataParallelModel(_model(), 0)
    opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 4), 4, 1)
    losses = []
    for s_ in range(2):
        x = torch.from_numpy(synth.make_clips(BS, FS, 224, seed=21 + s_)).cuda()
        loss = model(x, None, True, False)
        opt.step(loss)
        model.normalize_prototypes()
        losses.append(float(loss.item()))
    params = dict(model.get_non_ddp_model().named_parameters())
    ret["out"] = dict(losses=losses, calls=dict(calls), params={n: params[n].detach().cpu().numpy() for n in WATCH})
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_calls_execute_on_one_gpu():
    """The nccl / RCCL code path on the 1-GPU box: a one-rank communicator carries the score all-gather and the three gradient
    buckets of every step (asynchronously, on RCCL's streams); two optimizer steps must reproduce the plain single-process run
    bit for bit (mean over one rank = identity)."""
    import torch.multiprocessing as mp

    from timetuning_amd import synth
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer

    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    p = ctx.Process(target=_single_rank_nccl_worker, args=(_free_port(), ret))
    p.start()
    p.join(500)
    assert p.exitcode == 0, p.exitcode
    out = ret["out"]
    assert out["calls"]["all_gather"] == 2 and out["calls"]["all_reduce"] == 8, out["calls"]   # per step: 1 gather + 4 buckets

    model = _model()
    opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 4), 4, 1)
    for s_ in range(2):
        x = torch.from_numpy(synth.make_clips(BS, FS, 224, seed=21 + s_)).cuda()
        loss = model.get_loss(x)
        opt.step(loss)
        model.normalize_prototypes()
        assert loss.item() == out["losses"][s_]
    params = dict(model.named_parameters())
    for n in WATCH:
        assert np.array_equal(params[n].detach().cpu().numpy(), out["params"][n]), n


@pytest.mark.timeout(900)
def test_bench_gpus_2_through_its_own_launcher():
    """`python bench.py --gpus 2` with no external launcher: the parent starts two ranks (here both on cuda:0, over gloo,
    because the box has one GPU), rank 0 prints ONE line with n_gpus 2, twice the global batch and the process-group record."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, TT_BENCH_SHARE_DEVICE="1", TT_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch_size", "2",
                        "--num_frames", "2", "--num_clusters", "50", "--no_cpu_baseline", "--no_alt_precision"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    rc = out["rccl"]
    assert rc["world_size"] == 2 and rc["backend"] == "gloo"
    # the exchange decided for itself before the warm-up (engine.autotune_exchange): three configurations timed, one kept, recorded
    au = rc["exchange_autotune"]
    assert au["world_size"] == 2 and "allgather/4" in au["ms_per_step"] and len(au["ms_per_step"]) == 3 and all(v > 0 for v in au["ms_per_step"].values())
    assert au["sinkhorn_exchange"] in ("allgather", "allreduce") and au["grad_buckets"] in (1, 4)
    # the exchange of the instrumented step under that choice: ONE all-gather of the score rows (2 ranks x 2 clips x 196 patches x 50
    # prototypes, fp32) or ten K-float all-reduces of the row sums; four gradient buckets or one - every trainable float exactly once -
    # each with the compute stream's exposed wait
    kinds = [w["collective"] for w in rc["waits"]]
    if au["sinkhorn_exchange"] == "allgather":
        assert kinds[0] == "all_gather(scores)" and rc["waits"][0]["bytes"] == 2 * 2 * 196 * 50 * 4
        n_sk = 1
    else:
        assert kinds[:10] == ["all_reduce(sinkhorn row sums)"] * 10 and all(w["bytes"] == 50 * 4 for w in rc["waits"][:10])
        n_sk = 10
    buckets = [w for w in rc["waits"] if w["collective"].startswith("all_reduce(grad bucket")]
    assert len(buckets) == au["grad_buckets"] and rc["collectives_per_step"] == n_sk + au["grad_buckets"]
    assert sum(w["bytes"] for w in buckets) == 4 * (5_700_096 - 200 * 256 + 50 * 256)          # 5.70 M trainable floats at K = 200
    assert all(w["exposed_wait_ms"] >= 0 for w in rc["waits"]) and rc["exposed_wait_ms"] >= 0 and rc["bytes"] > 0
    assert out["value"] > 0 and out["scaling"] == "weak" and np.isfinite(out["loss"])


@pytest.mark.timeout(900)
def test_training_cli_spawns_two_ranks(tmp_path):
    """`python -m timetuning_amd.time_tuning -g 2` (mp.spawn, time_tuning.py:714-717) end to end on synthetic clips: two ranks
    (both on cuda:0 over gloo here), teacher + queue, the rank-0 evaluation at epoch 0 with the barrier behind it
    (:634-648), per-rank queues of queue_size // world_size rows, a checkpoint with the reference's DDP key layout."""
    import subprocess
    import sys

    env = dict(os.environ, TT_SHARE_DEVICE="1", TT_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "timetuning_amd.time_tuning", "-g", "2", "--dataset", "synthetic", "--model_path", "",
                        "--batch_size", "2", "--num_frames", "2", "--num_clusters", "20", "--num_epochs", "1", "--steps_per_epoch", "2",
                        "--use_queue", "1", "--queue_size", "256", "--eval_clips", "4", "--logging_directory", str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=800, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Scores/localization" in r.stdout and r.stdout.count("Iteration:") == 2           # rank 0 only prints
    ck = torch.load(str(tmp_path / "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert all(k.startswith("model.module.") for k in ck["model"]) and "model.module.teacher_prototypes" in ck["model"]


@pytest.mark.timeout(900)
def test_one_rank_exchange_path_equals_no_exchange():
    """Round 6: the exchange path as an N-GPU rank runs it - a one-rank RCCL communicator (TT_EXCHANGE_SINGLE_RANK=1), the side streams
    of engine.TWO_STREAMS with each bucket's all-reduce issued from the weight-gradient stream, the double-buffered gradient arena -
    trains exactly what the plain one-process step trains (W = 1: the mean over ranks is the identity): four steps' losses and every
    parameter bit for bit; the gradients are views of the arena's current buffer (no clone per step); and accumulating over backward
    calls WITHOUT zero_grad stays correct across the buffer swap (g0 + g1, then + g1 again when the first buffer comes round).  In a
    child process: the communicator is process-global."""
    import json
    import subprocess
    import sys

    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_exchange_single_rank_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=800)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    o = json.loads(lines[-1][7:])
    assert o["losses_equal"] and o["params_equal"], o
    assert o["two_buffers"] and o["grads_in_current_buffer"], o
    assert o["accumulate_err"] < 1e-6 and o["accumulate3_err"] < 1e-6, o
