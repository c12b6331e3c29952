#!/usr/bin/env python3
"""Benchmark of the TimeTuning training hot path on MI355X (driver contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no external launcher (no WORLD_SIZE in the environment) the script starts the N rank processes itself, one
per GPU, over RCCL (``launch_ranks``); under ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` it is
one of the ranks.  Either way rank 0 prints the line, with ``n_gpus: N`` and ``rccl: {world_size, backend}``.

A step = one full training iteration of BASELINE.json configs[1] (C2): ViT-S/16, 4-frame 224x224 clips,
32 clips per GPU, 200 prototypes, no queue / teacher - ``TimeT.get_loss`` forward + backward, AdamW step,
prototype renormalisation.  Synthetic clips and random-init weights (portable generator), resident in HBM before the
timed region.  Prints ONE JSON line on rank 0:
  metric/value   clip-frames/sec, whole job (N GPUs x 32 clips x 4 frames per step), weak scaling
  dtype          the arithmetic of the timed step.  Default (round 4): "f32-split(f16x3)" - every nn.Linear and attention product of the
                 step on fp16-pair operands, three fp16 MFMAs per term into fp32 accumulators, per-op error at or under the exact-f32
                 MFMA kernels' own (tests/test_hip_pairs.py) and every golden / oracle test at unchanged fp32 tolerances; the exact-f32
                 MFMA step ("f32", rounds 1-3's headline) is reported beside it under ``alt_precision`` with its own roofline block.
  roofline       the dominant GEMM kernel (largest time share): algorithmic flops of its launches in one instrumented step / their
                 HIP-event durations, against the peak of the arithmetic it runs in (2.5 PFLOP/s dense fp16 MFMA / 3 products per term
                 = 833 TFLOP/s for the pair kernels; 157.3 TFLOP/s for the f32 MFMA kernels)
  cpu_baseline   the CPU oracle (reference-faithful structure) timed on this host's cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import timetuning_amd  # noqa: E402,F401  (first: it switches ROCm's hipGraph packet capture off before the runtime initialises)
import torch  # noqa: E402

F32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
TILE_NAMES = {0: "128x128", 1: "64x128", 2: "128x64", 3: "64x64"}
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md)
ALT_NOTES = {
    "f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32) on every matrix product: the headline of rounds 1-3",
    "f16x3": "fp32-accurate split mode: operands pre-split into fp16 pairs (hi, lo x 2^11), 3 fp16 MFMAs per product term (nominal peak "
             f"{BF16_MATRIX_PEAK_TFLOPS / 3:.0f} TFLOP/s); per-op error under the f32-MFMA kernels' own; same fp32 tolerances in the parity tests",
    "bf16x6": "fp32-accurate split mode: operands pre-split into 3 bf16 planes, 6 bf16 MFMAs per product term (nominal peak "
              f"{BF16_MATRIX_PEAK_TFLOPS / 6:.0f} TFLOP/s) on the blocks that keep no activations; same fp32 tolerances in the parity tests",
    "bf16x3": "2-plane split (~2^-16 per product), fp32 in HBM converted while staging",
    "bf16": "bf16 activations + weights in HBM, bf16 attention (BASELINE C4's MFMA bf16 path) on the blocks that keep no activations; "
            "not within the fp32 contract",
}


def kernel_source_sha16(files):
    """sha256 (first 16 hex digits) over the named source files under the repo root, in order; None when one is missing."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        path = os.path.join(REPO, f)
        if not os.path.isfile(path):
            return None
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16] if files else None


def kernel_label(name, tile):
    if name == "NT":
        return f"gemm_nt_fast_kernel<{TILE_NAMES[tile]}> (forward nn.Linear, whole tiles)"
    if name == "NTbf16":
        return f"gemm_nt_bf16_kernel<{TILE_NAMES[tile]}> (forward nn.Linear, fp32 operands converted while staged)"
    if name == "PAIRS8":
        return "gemm_pairs8s_kernel (nn.Linear on fp16-pair operands, persistent kernel, 3 MFMAs per term)"
    if name == "PAIRS_TN":
        return "gemm_pairs_tn_kernel (weight gradient from row pairs, transposing LDS reads) + fold"
    if name == "PAIRS":
        return "gemm_planes_kernel<PAIR> (nn.Linear on fp16-pair operands, general kernel)"
    if name.startswith("PLANES8_"):
        return f"gemm_planes8_kernel<P={name[8:]}> (forward nn.Linear on bf16-plane operands, persistent 8-phase kernel)"
    if name.startswith("PLANES"):
        return f"gemm_planes_kernel<P={name[6:]}> (forward nn.Linear on bf16-plane operands)"
    if name == "BWD":
        return "gemm_bwd_fused_kernel: dgrad + wgrad of one nn.Linear in one launch (+ split-K fold)"
    fam = {"NN": "dgrad (dy @ w)", "TN": "wgrad (dy^T @ x, split-K)", "patch": "patch embed", "NTgen": "forward nn.Linear, general kernel"}.get(name, name)
    return f"gemm f32 {TILE_NAMES[tile]}: {fam}"


def build_model(arch, K, device, teacher=False, queue=0, world=1):
    from timetuning_amd import synth
    from timetuning_amd.models import DistributedDataParallelModel, FeatureExtractor
    from timetuning_amd.time_tuning import TimeT

    fe = FeatureExtractor(arch, "", [1024, 1024, 512, 256], unfreeze_layers=["blocks.11", "blocks.10"], init="dino",
                          return_attention=False)
    model = TimeT(fe, K, prototype_init=torch.from_numpy(synth.make_prototypes(K, fe.feature_dim))).to(device)
    if world > 1:
        model = DistributedDataParallelModel(model, device.index)
    return model


USE_MASK = False  # --use_mask: the attention-masked variant of the objective (not the headline workload)


def train_step(model, opt, x, use_teacher):
    loss = model(x, None, True, USE_MASK)
    # optimizer.step(loss) + normalize_prototypes() + update_momentum_teacher(step) (time_tuning.py:659-663), one C call
    model.train_update(opt, loss, min(opt.global_step + 1, len(model.momentum_schedule) - 1) if use_teacher else 0)
    return loss


def _core(model):
    return model.get_non_ddp_model() if hasattr(model, "get_non_ddp_model") else model


def _snapshot(model, opt):
    """Everything a training step changes: trainable + teacher tensors, optimizer / scheduler state, the queue, the host generator."""
    import copy

    core = _core(model)
    tensors = [p for p in core.parameters() if p.requires_grad]
    if core.teacher is not None:
        tensors += list(core.teacher.parameters()) + [core.teacher_prototypes]
    return dict(tensors=[(t, t.detach().clone()) for t in tensors], opt=copy.deepcopy(opt.optimizer.state_dict()),
                sched=copy.deepcopy(opt.lr_scheduler.state_dict()) if opt.lr_scheduler is not None else None, global_step=opt.global_step,
                probe=core.probe_state())


def _restore(model, opt, st):
    from timetuning_amd import hip_ops

    for t, saved in st["tensors"]:
        # through a raw-pointer kernel (t <- t * 0 + saved): an ATen copy_ would advance the version counters that are part of the step
        # graph's signature and send the next step down the eager path
        hip_ops.ema_update_(t.data.view(-1), saved.view(-1), 1.0)
    import copy

    opt.optimizer.load_state_dict(copy.deepcopy(st["opt"]))
    if st["sched"] is not None:
        opt.lr_scheduler.load_state_dict(copy.deepcopy(st["sched"]))
    opt.global_step = st["global_step"]
    _core(model).restore_probe_state(st["probe"])


def graph_check(model, opt, x, use_teacher):
    """is the REPLAYED step the eager step, here, on the state the timed region left behind?  One more replayed step
    and - from the same state - one eager step: loss, every gradient and every updated parameter must be equal bit for bit."""
    core = _core(model)
    st = _snapshot(model, opt)
    n_graphs = len(core._step_graphs)
    lg = train_step(model, opt, x, use_teacher)
    gg = [p.grad.clone() for p in core.parameters() if p.grad is not None]
    pg = [t.detach().clone() for t, _ in st["tensors"]]
    replayed = len(core._step_graphs) == n_graphs and n_graphs > 0   # (no new capture, no fall-back to the eager path: it WAS a replay)
    _restore(model, opt, st)
    from timetuning_amd import engine

    # (the reference: the eager step on ONE stream - a captured step runs on one, engine.two_streams; what the side streams change is the
    # K-split rounding of some launches' left-over tiles, tests/test_hip_timet.py::test_two_streams_equal_one_stream)
    core._step_graph_on, keep_streams, engine.TWO_STREAMS = False, engine.TWO_STREAMS, False
    try:
        le = train_step(model, opt, x, use_teacher)
    finally:
        core._step_graph_on, engine.TWO_STREAMS = True, keep_streams
    ge = [p.grad.clone() for p in core.parameters() if p.grad is not None]
    torch.cuda.synchronize()
    ok = (replayed and float(lg.item()) == float(le.item()) and len(gg) == len(ge) and all(torch.equal(a, b) for a, b in zip(gg, ge))
          and all(torch.equal(a, t.detach()) for a, (t, _) in zip(pg, st["tensors"])))
    return {"replay_equals_eager": bool(ok), "was_replay": bool(replayed), "loss_replay": float(lg.item()), "loss_eager_ref": float(le.item()),
            "compared": f"loss, {len(ge)} gradients, {len(pg)} updated tensors, bit for bit, from the state the timed region left"}


DTYPE_NAMES = {"f16x3": "f32-split(f16x3)", "bf16x6": "f32-split(bf16x6)"}   # (the other modes are named by their mode string)


def instrumented_step(model, opt, x, use_teacher):
    """One more identical step with every GEMM launch bracketed by HIP events on the launch stream.  Returns ({(kernel, tile): [launches,
    flops, seconds]}, [the compute stream's waits on collectives in THAT step])."""
    from timetuning_amd import hip_ops

    from timetuning_amd import engine

    rec = []
    hip_ops.PROFILE = rec
    engine.RCCL_PROFILE = []
    core = model.get_non_ddp_model() if hasattr(model, "get_non_ddp_model") else model
    graph_on, core._step_graph_on = getattr(core, "_step_graph_on", False), False   # (launch by launch: a graph replay has no per-launch events)
    try:
        train_step(model, opt, x, use_teacher)
        torch.cuda.synchronize()
    finally:
        core._step_graph_on = graph_on
        hip_ops.PROFILE = None
        waits, engine.RCCL_PROFILE = engine.RCCL_PROFILE, None
    rccl = [{"collective": k, "bytes": n, "exposed_wait_ms": round(e0.elapsed_time(e1), 4)} for k, n, e0, e1 in waits]
    by = {}
    for name, tile, flops, e0, e1 in rec:
        d = by.setdefault((name, tile), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += flops
        d[2] += e0.elapsed_time(e1) * 1e-3
    return by, rccl


def by_label(prof, step_seconds):
    """{kernel label: launches, TFLOP/s, share of the step}, largest share first; instantiations that share a label (the plane
    kernels' tile variants) are summed."""
    acc = {}
    for (nm, tl), (c_, f_, s_) in prof.items():
        d = acc.setdefault(kernel_label(nm, tl), [0, 0.0, 0.0])
        d[0] += c_
        d[1] += f_
        d[2] += s_
    return {k: {"launches": c_, "tflops": round(f_ / s_ / 1e12, 1), "share_of_step": round(s_ / step_seconds, 3)}
            for k, (c_, f_, s_) in sorted(acc.items(), key=lambda kv: -kv[1][2])}


def rccl_report(dist, waits):
    """What carried the exchange and what it cost the compute stream in the instrumented step: per collective its payload and the
    time the compute stream sat in ``work.wait()`` (HIP events around the wait: 0 when RCCL had finished under the backward)."""
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "collectives_per_step": len(waits),
            "bytes": sum(w["bytes"] for w in waits), "exposed_wait_ms": round(sum(w["exposed_wait_ms"] for w in waits), 4), "waits": waits}


def single_rank_exchange_probe_child(a):
    """Child process of the 1-GPU bench (``--exchange_probe_child``): joins a ONE-rank ``nccl`` process group and runs the configured step
    with TT_EXCHANGE_SINGLE_RANK=1, i.e. with the very RCCL calls of an N-GPU run (1 asynchronous all-gather of the score rows + the
    gradient buckets' asynchronous all-reduces, engine.exchange_group) on a one-rank communicator - RCCL refuses two ranks on one
    device, so this is how a 1-GPU box exercises the exchange and the ``rccl`` fields an 8-GPU line will carry.  Prints one JSON object."""
    import torch.distributed as dist

    from timetuning_amd import hip_ops, synth
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ["TT_EXCHANGE_SINGLE_RANK"] = "1"
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0, device_id=device)
    bs, fs, K = a.batch_size, a.num_frames, a.num_clusters
    model = build_model(a.architecture, K, device)
    opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 64), 64, 1)
    if a.use_teacher:
        model.init_momentum_teacher()
        model.set_momentum_teacher_schedular_params(0.995, 1.0, 1, 64)
    if a.use_queue:
        model.init_queue(a.queue_size)
        model.set_queue(torch.nn.functional.normalize(torch.randn_like(model.queue), dim=1))
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1)).to(device)
    hip_ops.set_gemm_precision(a.precision)
    for _ in range(3):
        train_step(model, opt, x, a.use_teacher)
    torch.cuda.synchronize()
    _, waits = instrumented_step(model, opt, x, a.use_teacher)
    out = rccl_report(dist, waits)
    for _ in range(2):   # (back on the timed path: the instrumented step ran launch by launch on one stream)
        train_step(model, opt, x, a.use_teacher)
    torch.cuda.synchronize()
    n_t = max(5, a.steps)
    t0 = time.perf_counter()
    for _ in range(n_t):
        train_step(model, opt, x, a.use_teacher)
    torch.cuda.synchronize()
    out["ms_per_step_with_exchange"] = round((time.perf_counter() - t0) / n_t * 1e3, 3)
    out["note"] = "one-rank communicator on one GPU (TT_EXCHANGE_SINGLE_RANK=1): same calls, no peer; separate process, not part of `value`"
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


def single_rank_exchange_probe(argv, timeout_s=240):
    """Runs ``single_rank_exchange_probe_child`` in a fresh child process (a fresh interpreter started with subprocess - never a re-exec of this
    process, which has initialised the GPU) and returns its JSON object; a child that crashes, hangs or cannot initialise RCCL costs an
    ``error`` entry, never the bench line."""
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), *argv, "--exchange_probe_child"], env=env, capture_output=True, text=True,
                           timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"probe child exceeded {timeout_s} s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"probe child exited with {r.returncode}: {r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else 'no output'}"}
    return json.loads(lines[-1])


PEAK_BY_KERNEL = {"PLANES1": BF16_MATRIX_PEAK_TFLOPS, "PLANES2": BF16_MATRIX_PEAK_TFLOPS / 3, "PLANES3": BF16_MATRIX_PEAK_TFLOPS / 6,
                  "PLANES8_1": BF16_MATRIX_PEAK_TFLOPS, "PLANES8_3": BF16_MATRIX_PEAK_TFLOPS / 6,
                  "PAIRS8": BF16_MATRIX_PEAK_TFLOPS / 3, "PAIRS": BF16_MATRIX_PEAK_TFLOPS / 3,
                  "PAIRS_TN": BF16_MATRIX_PEAK_TFLOPS / 3}


def roofline_block(prof, step_seconds, precision):
    """The `roofline` object of a line: the GEMM kernel with the largest time share of the instrumented step - keyed by kernel LABEL, the
    same keying as ``by_kernel`` (tile variants of one kernel are one entry in both) - its algorithmic flops / its HIP-event durations,
    against the peak of the arithmetic it runs in."""
    acc = {}
    for (nm, tl), (c_, f_, s_) in prof.items():
        d = acc.setdefault(kernel_label(nm, tl), [0, 0.0, 0.0, nm])
        d[0] += c_
        d[1] += f_
        d[2] += s_
    dom_label, (cnt, flops, sec, dom_name) = max(acc.items(), key=lambda kv: kv[1][2])
    all_flops = sum(v[1] for v in prof.values())
    all_sec = sum(v[2] for v in prof.values())
    # peak the dominant kernel is priced against: dense f32 MFMA, or dense 16-bit MFMA / the MFMAs it issues per product term
    peak = PEAK_BY_KERNEL.get(dom_name, BF16_MATRIX_PEAK_TFLOPS / 3 if (dom_name == "NTbf16" and precision == "bf16x3") else
                              BF16_MATRIX_PEAK_TFLOPS if dom_name == "NTbf16" else F32_MATRIX_PEAK_TFLOPS)
    # HBM bytes per launch of the dominant kernel from the committed PMC pass (tools/pmc_traffic.py) - only when that pass
    # measured THIS kernel
    traffic = traffic_git_head = None
    tpath = os.path.join(REPO, "profiles", "dominant_kernel_traffic.json")
    if os.path.isfile(tpath):
        tj = json.load(open(tpath))
        # ... and only while the kernel's source is the one that pass ran (the snapshot on the GPU box has no .git: the tie is a hash of
        # the source files, written by tools/pmc_traffic.py next to the commit it was taken at)
        if tj.get("kernel_label", "gemm_nt_fast_kernel<64x128>") in dom_label and tj.get("kernel_source_sha16") == kernel_source_sha16(tj.get("kernel_sources", [])):
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_git_head = tj.get("git_head")
    return {"bound": "mfma", "precision": DTYPE_NAMES.get(precision, precision),
            "kernel": dom_label, "launches_per_step": cnt,
            "achieved": round(flops / sec / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(flops / sec / 1e12 / peak, 4), "traffic": traffic,
            # NOT a counter of this run: HBM bytes per launch of this kernel from the committed PMC passes
            "traffic_source": None if traffic is None else "profiles/dominant_kernel_traffic.json (rocprofv3 --pmc, separate passes)",
            "traffic_git_head": traffic_git_head,
            "avg_launch_us": round(sec / cnt * 1e6, 2),
            "all_gemm_tflops": round(all_flops / all_sec / 1e12, 2),
            "gemm_share_of_step": round(all_sec / step_seconds, 3),
            # every GEMM family of the step (forward Linears, dgrad / wgrad, patch embed, plane / pair kernels): launches,
            # achieved TFLOP/s on algorithmic flops, share of the step - the dominant one is the roofline kernel above
            "by_kernel": by_label(prof, step_seconds)}


def sinkhorn_rate(device, B=6272, K=200, iters=10, reps=30):
    from timetuning_amd import hip_ops, synth

    x = torch.nn.functional.normalize(torch.from_numpy(synth.normal("bench.sk.x", (B, 256))), dim=1)
    p = torch.from_numpy(synth.make_prototypes(K, 256))
    scores = (x @ p.t()).to(device)
    for _ in range(3):
        hip_ops.sinkhorn(scores, iters)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip_ops.sinkhorn(scores, iters)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    algo_bytes = 4.0 * K * B * (2 * iters + 2)  # two sweeps of Q per iteration
    swept_bytes = 4.0 * K * B * (iters + 3)     # what tt_sinkhorn moves: scaling-vector form, ONE sweep per iteration (13 for 10 iterations)
    return iters / sec, algo_bytes / sec / 1e9, swept_bytes / sec / 1e9


def cpu_baseline(fs, K, budget_s=30.0, arch="dino-s16", precision="f32"):
    """The oracle in the reference's own structure (4 backbone passes per frame, per-sample host propagation) on a
    bounded sample of the timed step: bs=2 clips instead of the full batch (per-clip work is identical).  Also the parity of the
    GPU path IN THE TIMED PRECISION MODE against this fp32 oracle on that sample (patch embeddings, assignment logits)."""
    from oracle import timet_oracle as O
    from timetuning_amd import synth

    # torch-CPU stops scaling at 16 threads on these op sizes and then collapses (tools/cpu_threads_sweep.py on the 256-core
    # box, profiles/r02_cpu_threads_sweep.txt: 14.3 / 17.4 / 10.7 / 4.3 / 1.7 / 0.04 clip-frames/s at 8 / 16 / 32 / 64 / 128 / 256
    # threads): the baseline runs at its best setting
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 16)
    torch.set_num_threads(cores)
    bs = 8   # clips per CPU step (round 2 ran 2: too little parallel work per op for a fair host number)
    om = O.build_oracle(arch, K, (1024, 1024, 512, 256), mode="dino")
    opt = O.SwavOptimizerOracle(om, 1e-5, 1e-4, O.cosine_scheduler(0.04, 0.4, 1, 128), 128, 1)
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1))
    # parity of the two paths on this very sample, before either is trained (same portable weights): patch embeddings
    # (head outputs) and assignment logits (patch x prototype scores) - the north-star quantities, bound 1e-3 relative
    with torch.no_grad():
        flat = x.view(bs * fs, 3, 224, 224)
        of, _ = om.feature_extractor(flat, faithful=False)
        osc = om.get_feature_prototype_similarity(of.reshape(-1, of.shape[-1]))
        from timetuning_amd import hip_ops

        gm = build_model(arch, K, torch.device("cuda", 0))
        before = hip_ops.get_gemm_precision()
        hip_ops.set_gemm_precision(precision)
        try:
            gf, _ = gm.feature_extractor(flat.cuda())
            gsc = gm.get_feature_prototype_similarity(gf.reshape(-1, gf.shape[-1]))
        finally:
            hip_ops.set_gemm_precision(before)
        rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        parity = {"patch_embeddings_rel_err": rel(gf.cpu(), of), "assignment_logits_rel_err": rel(gsc.cpu(), osc), "gpu_precision": DTYPE_NAMES.get(precision, precision),
                  "bound": 1e-3 if precision in ("f32", "f16x3", "bf16x6", "bf16x3") else None}
        del gm, gf, gsc
        # the second half of the metric on the host: the reference's Sinkhorn (my_utils.py:246-274) at the C2 shape
        sk_in = torch.exp(torch.nn.functional.normalize(torch.randn(6272, 256), dim=1) @ torch.nn.functional.normalize(torch.randn(K, 256), dim=1).t() / 0.05).t()
        O.sinkhorn(sk_in.clone(), 10)
        t0 = time.perf_counter()
        for _ in range(3):
            O.sinkhorn(sk_in.clone(), 10)
        sk_cpu = 30.0 / (time.perf_counter() - t0)
    n, t_total = 0, 0.0
    while n == 0 or (t_total < 0.5 * budget_s and n < 64):  # ~15-25 s of CPU work, at least one step
        t0 = time.perf_counter()
        loss = om.get_loss(x, faithful=True, mask_features=USE_MASK)
        opt.zero_grad()
        loss.backward()
        opt.step()
        om.normalize_prototypes()
        t_total += time.perf_counter() - t0
        n += 1
    return {"value": round(bs * fs * n / t_total, 4), "unit": "clip-frames/sec", "cores": cores, "threads": cores, "host_cores": host_cores, "kind": "port",
            "threads_note": "torch-CPU peaks at 16 threads on these op sizes and collapses beyond (profiles/r02_cpu_threads_sweep.txt); "
                            "a weak baseline by construction - it is reported, never the target",
            "sample": f"{n} training step(s) of {bs} clips x {fs} frames of {arch} (per-clip work identical to the timed batch), torch-CPU fp32, "
                      "reference-faithful structure (4 ViT passes per frame, per-sample host label propagation)",
            "seconds_per_step": round(t_total / n, 3), "sinkhorn_iters_per_sec": round(sk_cpu, 1), "parity_vs_gpu": parity}


def _free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv) -> int:
    """``python bench.py --gpus N`` without an external launcher: start N rank processes of this script (one per GPU, the
    reference's ``mp.spawn(time_tuning, nprocs=args.gpus)``, time_tuning.py:714-717) and wait for them.  The parent never touches
    the GPU - no ``torch.cuda`` / HIP call happens before or after the children are started - and the children are fresh
    interpreters (``subprocess``, no fork of an initialised runtime, no re-exec).  Rank 0's JSON line passes straight through
    on the inherited stdout.  Returns the exit code (non-zero as soon as any rank fails; the others are then terminated so
    that nobody hangs in a collective)."""
    import signal
    import subprocess

    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, TT_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank process {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in live:
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.send_signal(signal.SIGTERM)
        rc = 130
    for q in procs:
        try:
            q.wait(timeout=30)
        except subprocess.TimeoutExpired:
            q.kill()
    return rc


def dry_rank(a, world: int, rank: int) -> None:
    """TT_BENCH_DRY=1 (launcher self-test, runs without a GPU): the rank joins the process group, proves through an all-gather
    that ``world`` distinct ranks are present and rank 0 prints a JSON line in the bench's shape - no kernels, no timing."""
    import torch.distributed as dist

    if os.environ.get("TT_BENCH_DRY_FAIL_RANK") == str(rank):  # launcher self-test: a rank that dies must fail the whole run
        sys.exit(3)
    dist.init_process_group(backend=os.environ.get("TT_BENCH_BACKEND", "gloo"), init_method="env://", world_size=world, rank=rank)
    seen = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seen, torch.tensor([rank, os.getpid()], dtype=torch.int64))
    if rank == 0:
        print(json.dumps({"metric": "clip-frames/sec", "value": None, "dry": True, "n_gpus": world,
                          "ranks": [int(t[0]) for t in seen], "pids": [int(t[1]) for t in seen],
                          "config": {"global_batch": a.batch_size * world, "parallelism": f"dp{world}"},
                          "rccl": {"world_size": dist.get_world_size(), "backend": dist.get_backend()}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=32, help="clips per GPU (C2: 32)")
    ap.add_argument("--num_frames", type=int, default=4)
    ap.add_argument("--num_clusters", type=int, default=200)
    ap.add_argument("--architecture", default="dino-s16")
    ap.add_argument("--use_teacher", action="store_true")
    ap.add_argument("--use_queue", action="store_true")
    ap.add_argument("--queue_size", type=int, default=16384, help="global queue rows (the reference's default); each rank holds queue_size // world")
    ap.add_argument("--use_mask", action="store_true", help="time the --use_mask variant (attention foreground masks) instead")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32", "bf16x6", "bf16x3", "bf16"],
                    help="arithmetic of the matrix products of the TIMED region (default f16x3 = the fp32-accurate split mode: per-op error "
                         "under the f32-MFMA kernels' own, every parity test at fp32 tolerances; f32 = exact fp32 MFMA; "
                         "hip_ops.set_gemm_precision documents the others)")
    ap.add_argument("--no_alt_precision", action="store_true", help="skip the secondary measurements in the other precision modes")
    ap.add_argument("--no_exchange_probe", action="store_true", help="skip the one-rank RCCL probe of the exchange path (1-GPU runs)")
    ap.add_argument("--step_graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step's launch sequence as ONE captured hipGraph (TimeT.enable_step_graph, the driver's default too): "
                         "auto = one GPU AND a launch-bound step (at most 10 k token rows: C1 2.0 against 3.5 ms; C2 - C5 are equal either way "
                         "and stay launch by launch), off with N > 1")
    ap.add_argument("--no_exchange_autotune", action="store_true",
                    help="N > 1: keep the default exchange (all-gather Sinkhorn, 4 gradient buckets) instead of timing the variants first")
    ap.add_argument("--exchange_probe_child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    global USE_MASK
    USE_MASK = a.use_mask

    if a.exchange_probe_child:
        return single_rank_exchange_probe_child(a)

    # ---- N > 1 without an external launcher: this process only starts the N rank processes (before any GPU call)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    import torch.distributed as dist

    from timetuning_amd import synth
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    if os.environ.get("TT_BENCH_DRY"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        return dry_rank(a, world, rank)
    if os.environ.get("TT_BENCH_SHARE_DEVICE"):  # test aid: all ranks on cuda:0 of a 1-GPU box (use with TT_BENCH_BACKEND=gloo)
        local = 0
    elif world > 1 and torch.cuda.device_count() < world:  # (device_count does not initialise the runtime)
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        backend = os.environ.get("TT_BENCH_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local if world > 1 else 0)

    # the queue's permutations (torch.randperm from the CPU generator, time_tuning.py:259 of the reference) and the queue's initial rows come
    # from torch's generators: seeded, so that a configuration's `loss` repeats from run to run (round 6: C3's wandered by +- 0.1 unseeded)
    torch.manual_seed(20 + rank)
    bs, fs, K = a.batch_size, a.num_frames, a.num_clusters
    model = build_model(a.architecture, K, device, world=world)
    total_steps = a.steps + a.warmup + 200
    opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, total_steps), total_steps, 1)
    if a.use_teacher:
        model.init_momentum_teacher()
        model.set_momentum_teacher_schedular_params(0.995, 1.0, 1, total_steps)
    if a.use_queue:
        model.init_queue(a.queue_size // world)
        model.set_queue(torch.nn.functional.normalize(torch.randn_like(model.queue), dim=1))
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1 + rank)).to(device)  # resident in HBM before timing
    step_graph = world == 1 and a.step_graph in ("on", "auto")
    if step_graph and a.step_graph == "auto" and not timetuning_amd.step_graph_safe():
        step_graph = False   # (the runtime's hipGraph packet capture could not be switched off in this process: launches only; "on" raises)
    if step_graph:
        from timetuning_amd.time_tuning import TimeT

        # auto: replay launch-bound steps only (TimeT.STEP_GRAPH_AUTO_MAX_ROWS: C1-sized steps; C2 - C5 run launch by launch, as an N > 1 rank does)
        model.enable_step_graph(max_token_rows=TimeT.STEP_GRAPH_AUTO_MAX_ROWS if a.step_graph == "auto" else None)
        rows = bs * fs * (1 + model.feature_extractor.spatial_resolution ** 2)
        step_graph = a.step_graph == "on" or rows <= TimeT.STEP_GRAPH_AUTO_MAX_ROWS

    from timetuning_amd import hip_ops

    hip_ops.set_gemm_precision(a.precision)
    from timetuning_amd import engine

    if world > 1 and not a.no_exchange_autotune:
        # which Sinkhorn exchange / gradient bucketing is faster on THIS node's links is measured, once, before the warm-up (outside the
        # timed region): forward + backward of the real step, no update (engine.autotune_exchange; the choice is printed under `rccl`)
        def _probe_step():
            model.zero_grad(set_to_none=True)
            model(x, None, True, USE_MASK).backward()

        core_ = model.get_non_ddp_model() if hasattr(model, "get_non_ddp_model") else model
        engine.autotune_exchange(_probe_step, device, log=(lambda m: print(m, file=sys.stderr, flush=True)),
                                 state=(core_.probe_state, core_.restore_probe_state))   # queue + host generator put back: same workload for every variant
        model.zero_grad(set_to_none=True)
    if step_graph:
        # the graph of a step is captured at the SECOND occurrence of its signature (the first runs eagerly); with a queue the signature
        # changes once, when the queue fills: run until the replay is steady, so that no capture lands in the W warm-up or the K timed steps
        # whatever W is (these steps are extra warm-up, untimed)
        prime = 2
        if a.use_queue:
            per_step = min(bs * 10, model.queue.shape[0])
            prime += (model.queue.shape[0] + per_step - 1) // per_step + 2
        for _ in range(prime):
            train_step(model, opt, x, a.use_teacher)
    for _ in range(a.warmup):
        train_step(model, opt, x, a.use_teacher)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = train_step(model, opt, x, a.use_teacher)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    # the replayed step against the eager one (same state), and the eager step's own time beside the replayed one's
    check = None
    eager_ms = None
    if step_graph:
        check = graph_check(model, opt, x, a.use_teacher)
        if not check["replay_equals_eager"]:
            raise SystemExit(f"bench.py: the replayed step is NOT the eager step: {check}")
        model._step_graph_on = False
        for _ in range(2):
            train_step(model, opt, x, a.use_teacher)
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(a.steps):
            train_step(model, opt, x, a.use_teacher)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - te) / a.steps * 1e3
        model._step_graph_on = True

    out = None
    # the instrumented step contains the step's collectives (score all-gather, gradient all-reduce): EVERY rank runs it
    prof, headline_waits = instrumented_step(model, opt, x, a.use_teacher)
    # secondary, clearly-labelled measurements of the other arithmetic modes (same step); `value` above is always the --precision
    # mode.  Every rank takes part (collectives).  The exact-f32 line and the other fp32-accurate split carry their own roofline blocks.
    alt = {}
    if a.precision in ("f16x3", "f32") and not a.no_alt_precision:
        for mode in [m for m in ("f32", "f16x3", "bf16x6", "bf16") if m != a.precision]:
            hip_ops.set_gemm_precision(mode)
            for _ in range(2):
                train_step(model, opt, x, a.use_teacher)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            ta = time.perf_counter()
            for _ in range(5):
                la = train_step(model, opt, x, a.use_teacher)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - ta
            if world > 1:
                tt_ = torch.tensor([dt], device=device, dtype=torch.float64)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                dt = float(tt_.item())
            alt[mode] = {"clip_frames_per_sec": round(world * bs * fs * 5 / dt, 1), "ms_per_step": round(dt / 5 * 1e3, 3), "loss": round(float(la.item()), 5),
                         "dtype": DTYPE_NAMES.get(mode, mode), "note": ALT_NOTES[mode]}
            if mode in ("f32", "f16x3", "bf16x6"):
                aprof, _ = instrumented_step(model, opt, x, a.use_teacher)
                if rank == 0:
                    alt[mode]["roofline"] = roofline_block(aprof, dt / 5, mode)
        hip_ops.set_gemm_precision(a.precision)
    if rank == 0:
        roof = roofline_block(prof, elapsed / a.steps, a.precision)
        sk_rate, sk_gbs, sk_swept = sinkhorn_rate(device) if world == 1 else (None, None, None)
        workload = ("C2: " if (a.architecture, bs, fs, K, a.use_teacher, a.use_queue) == ("dino-s16", 32, 4, 200, False, False) else
                    "C3 (per-GPU share): " if (a.architecture, bs, fs, K, a.use_teacher, a.use_queue) == ("dino-s16", 32, 4, 200, True, True) else
                    "C4 (per-GPU share): " if (a.architecture, bs, fs, K) == ("dino-b16", 16, 8, 400) else
                    "C5 (per-GPU share): " if (a.architecture, bs, fs, K) == ("dino-s8", 16, 4, 200) else "")
        out = {
            "metric": "clip-frames/sec", "value": round(world * bs * fs * a.steps / elapsed, 2), "unit": "clip-frames/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAMES.get(a.precision, a.precision), "data": "synthetic",
            "config": {"workload": workload + f"{a.architecture} full TimeT training step (fwd+bwd+AdamW), {fs}-frame 224x224 clips, {bs} clips/GPU, "
                                   f"{K} prototypes" + (", EMA teacher" if a.use_teacher else "") + (f", queue {a.queue_size // world} rows/rank" if a.use_queue else "") +
                                   (", use_mask" if a.use_mask else ""),
                       "clips_per_gpu": bs, "num_frames": fs, "num_clusters": K,
                       "global_batch": bs * world, "parallelism": f"dp{world}"},
            "loss": round(final_loss, 5),
            "step_graph": bool(step_graph),   # True: the timed steps replay ONE captured hipGraph of the step's launch sequence (+ the eager optimizer call)
            "ms_per_step_eager": None if eager_ms is None else round(eager_ms, 3),   # the same step launch by launch (what an N > 1 rank runs), same process
            "graph_check": check,             # the replayed step == the eager step from the same state, bit for bit (asserted)
            # proof of what carried the exchange: RCCL ("nccl") saw this many ranks (None for the single-process run)
            # with the compute stream's exposed wait per collective in the instrumented step
            "rccl": dict(rccl_report(dist, headline_waits), exchange_autotune=engine.EXCHANGE_CHOICE) if world > 1 else None,
            "roofline": roof,
            "alt_precision": alt or None,
            "sinkhorn": None if sk_rate is None else {"iters_per_sec": round(sk_rate, 1), "algorithmic_GBps": round(sk_gbs, 1),
                                                      "swept_GBps": round(sk_swept, 1),
                                                      "note": "algorithmic = SURVEY 8(d)'s 4 K B (2 iters + 2) bytes; swept = the 13 sweeps the "
                                                              "scaling-vector kernel really makes (iters + 3); the 5 MB matrix is cache-resident",
                                                      "shape": "K=200 x B=6272, 10 iterations"},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fs, K, arch=a.architecture, precision=a.precision)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not a.no_exchange_probe:
            out["rccl_single_rank_probe"] = single_rank_exchange_probe([v for v in sys.argv[1:] if v != "--exchange_probe_child"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
