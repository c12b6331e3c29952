"""CPU-only checks: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares; the host
mirror of the reference interface (CLI, parameter groups, state_dict layout, schedules, queue bookkeeping) behaves as
the reference does; and the product path refuses to run without the HIP library / a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from timetuning_amd import _lib, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "timetuning_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    if not os.path.isfile(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/timetuning_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in timetuning_amd/_lib.py"
    assert set(_lib.SIGNATURES) <= set(declared)
    assert lib.tt_abi_version() == 8
    # size queries are pure host functions and can be called without a GPU
    assert lib.tt_sinkhorn_workspace_bytes(6272, 200) >= 6272 * 200 * 4
    assert lib.tt_ce_workspace_bytes(100) == 400
    assert lib.tt_gemm_tile_choice(25216, 1152, 1) in (0, 1, 2, 3)


def test_no_cpu_fallback():
    from timetuning_amd import hip_ops
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.time_tuning import TimeT

    with pytest.raises(_lib.HipLibraryError):
        hip_ops.linear_fwd(torch.zeros(4, 4), torch.zeros(4, 4))
    fe = FeatureExtractor("dino-s16", "", [128, 128, 64, 32], unfreeze_layers=["blocks.11", "blocks.10"], vit_cfg=synth.ARCHS["tiny-s16"])
    model = TimeT(fe, 20)
    with pytest.raises(_lib.HipLibraryError):
        model(torch.zeros(1, 2, 3, 224, 224), None, True, False)
    with pytest.raises(_lib.HipLibraryError):
        fe(torch.zeros(1, 3, 224, 224))


def test_oracle_is_not_imported_by_the_product():
    import ast

    pkg = os.path.join(REPO, "timetuning_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n.split(".")[0] == "oracle" for n in names), f"{fn} imports the oracle"


def test_cli_surface_and_bool_quirk():
    from timetuning_amd.time_tuning import build_parser

    a = build_parser().parse_args([])
    assert (a.architecture, a.num_clusters, a.num_frames, a.batch_size) == ("dino-s16", 200, 4, 128)
    assert a.use_teacher is True and a.use_queue is False and a.use_projection_head is True and a.use_mask is False
    assert (a.EMA_decay, a.queue_size, a.head_lr, a.lr_scheduler) == (0.995, 16384, 1e-4, "CosineAnnealingLR")
    # type=bool: any non-empty string is True (time_tuning.py:701; README.md:58 passes "--use_queue False")
    assert build_parser().parse_args(["--use_queue", "False"]).use_queue is True
    b = build_parser().parse_args(["-g", "8", "-n", "1", "-nr", "0", "--num_clusters", "400", "--use_teacher", ""])
    assert b.gpus == 8 and b.num_clusters == 400 and b.use_teacher is False


def test_param_groups_and_state_dict_layout(golden):
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer, TimeT

    g = golden("timet_tiny_tq")
    fe = FeatureExtractor("dino-s16", "", [int(v) for v in g["head_list"]], unfreeze_layers=["blocks.11", "blocks.10"],
                          vit_cfg=synth.ARCHS["tiny-s16"], init="stress")
    model = TimeT(fe, 20)
    model.init_momentum_teacher()
    model.init_queue(40)
    assert set(model.state_dict().keys()) == {str(k) for k in g["state_dict_keys"]}  # queue is not a buffer (time_tuning.py:107)
    opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, 4), 4, 1)
    assert [len(gr["params"]) for gr in opt.optimizer.param_groups] == list(g["group_sizes"]) == [1, 0, 4, 4, 8, 16]
    assert [gr["lr"] for gr in opt.optimizer.param_groups] == [1e-4, 1e-4, 1e-4, 1e-4, 1e-5, 1e-5]
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    frozen = [n for n, p in model.feature_extractor.backbone.named_parameters() if not p.requires_grad]
    assert "norm.weight" in frozen and "pos_embed" in frozen and "blocks.9.mlp.fc2.weight" in frozen
    assert trainable == sum(p.numel() for gr in opt.optimizer.param_groups for p in gr["params"])
    assert fe.trainable_block_ids() == [10, 11]
    assert all(not p.requires_grad for p in model.teacher.parameters())


def test_full_size_parameter_counts():
    """5,700,096 trainable parameters in 33 tensors for dino-s16 + head + 200 prototypes."""
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.time_tuning import TimeT

    fe = FeatureExtractor("dino-s16", "", [1024, 1024, 512, 256], unfreeze_layers=["blocks.11", "blocks.10"])
    model = TimeT(fe, 200)
    tr = [p for p in model.parameters() if p.requires_grad]
    assert len(tr) == 33 and sum(p.numel() for p in tr) == 5_700_096
    assert fe.feature_dim == 256 and fe.spatial_resolution == 14
    w = synth.make_vit_weights(mode="dino", **synth.ARCHS["dino-s16"])
    assert np.array_equal(fe.backbone.blocks[3].attn.qkv.weight.detach().numpy(), w["blocks.3.attn.qkv.weight"])


def test_planes_route_decisions_without_a_gpu():
    """tt_linear_fwd_planes_route is host logic (shape / epilogue eligibility and the tile decomposition of the persistent plane GEMM); with
    no device it assumes 256 CUs."""
    lib = _lib.load()
    route = lambda P, M, N, K, act=0, bias=1, res=0, y=1, po=0, pre=0: lib.tt_linear_fwd_planes_route(P, M, N, K, act, bias, res, y, po, pre)
    assert route(1, 25216, 2304, 768, y=0, po=1) == 8            # ViT-B/16 qkv, bf16 out
    assert route(1, 25216, 768, 3072, res=1) == 8                # fc2, fp32 + residual
    assert route(3, 25216, 1536, 384, act=1, y=0, po=3) == 8     # ViT-S/16 fc1 in the fp32-accurate mode
    assert route(1, 25216, 2304, 768, act=1) == 0                # GELU with an fp32 output is not a compiled epilogue
    assert route(1, 6304, 768, 768, res=1) == 0                  # 75 tiles: the small-tile kernel
    assert route(3, 25216, 1152, 200) == 0                       # K not a multiple of two K-tiles


def test_queue_fullness_is_tracked_on_the_host():
    """Own pushes are counted on the host; a write from outside (copy_, set_queue) is detected by the storage / version signature and
    answered by the reference's own check of the last row (time_tuning.py:207)."""
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.time_tuning import TimeT

    fe = FeatureExtractor("dino-s16", "", [128, 128, 64, 32], vit_cfg=synth.ARCHS["tiny-s16"])
    m = TimeT(fe, 20)
    assert not m.queue_is_full()
    m.init_queue(40)
    assert m.queue.shape == (40, 32) and not m.queue_is_full()
    m._queue_pushed(20)
    assert not m.queue_is_full()
    m._queue_pushed(20)
    assert m.queue_is_full()
    # a drop-in caller fills the tensor itself, as the reference's callers may
    m.init_queue(40)
    m.queue[:10].fill_(1.0)
    assert not m.queue_is_full()          # last row still zero
    m.queue.copy_(torch.ones(40, 32))
    assert m.queue_is_full()
    m.queue.zero_()
    assert not m.queue_is_full()
    m.set_queue(torch.ones(40, 32))
    assert m.queue_is_full()
    m.set_queue(torch.ones(24, 32))       # a different size re-allocates
    assert m.queue.shape == (24, 32) and m.queue_is_full()
    m.queue = torch.zeros(24, 32)         # a replaced tensor
    assert not m.queue_is_full()


def test_partial_unfreeze_follows_the_reference_substring_match():
    """models.py:929-935 accepts any substring; parts of a block train with the whole block on the backward path, tensors below the
    first block raise at construction."""
    from timetuning_amd.models import FeatureExtractor

    cfg = synth.ARCHS["tiny-s16"]
    fe = FeatureExtractor("dino-s16", "", [128, 128, 64, 32], unfreeze_layers=["blocks.11.attn", "blocks.9.mlp.fc2.weight"], vit_cfg=cfg)
    names = {n for n, p in fe.backbone.named_parameters() if p.requires_grad}
    assert names == {"blocks.11.attn.qkv.weight", "blocks.11.attn.qkv.bias", "blocks.11.attn.proj.weight", "blocks.11.attn.proj.bias",
                     "blocks.9.mlp.fc2.weight"}
    assert fe.trainable_block_ids() == [9, 11]
    for bad in (["patch_embed"], ["pos_embed"], ["cls_token"], ["blocks.11", "embed"]):
        with pytest.raises(NotImplementedError):
            FeatureExtractor("dino-s16", "", [128, 128, 64, 32], unfreeze_layers=bad, vit_cfg=cfg)


def test_unknown_architecture_raises_clearly():
    from timetuning_amd.models import FeatureExtractor

    with pytest.raises(ValueError, match="unknown architecture"):
        FeatureExtractor("resnet50", "")


def test_schedules_match_reference(golden):
    from timetuning_amd.my_utils import cosine_scheduler

    g = golden("schedules")
    np.testing.assert_allclose(cosine_scheduler(0.04, 0.4, 1, 4), g["wd_1_4"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(cosine_scheduler(0.995, 1.0, 2, 5), g["ema_2_5"], rtol=0, atol=1e-15)


def test_checkpoint_round_trip(tmp_path):
    """save_checkpoint / load_checkpoint (time_tuning.py:460-505): model (student, teacher, prototypes, queue-free state_dict),
    optimizer state, scheduler and global_step survive a round trip; a missing file resumes from epoch 0."""
    import torch

    from timetuning_amd import synth
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer, TimeT, load_checkpoint, save_checkpoint

    def make(seed):
        cfg = synth.ARCHS["tiny-s16"]
        fe = FeatureExtractor("dino-s16", "", [128, 128, 64, 32], unfreeze_layers=["blocks.11", "blocks.10"], vit_cfg=cfg, init="stress", seed=seed)
        model = TimeT(fe, 20, prototype_init=torch.from_numpy(synth.make_prototypes(20, 32, seed=seed)))
        model.init_momentum_teacher()
        opt = SwavOptimizer(model, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 2, 4), 4, 2)
        return model, opt

    a, oa = make(1)
    for p in a.parameters():           # optimizer state as after one step, without needing the GPU
        if p.requires_grad:
            oa.optimizer.state[p] = dict(step=torch.tensor(1.0), exp_avg=torch.full_like(p, 0.5), exp_avg_sq=torch.full_like(p, 0.25))
    oa.global_step = 3
    oa.lr_scheduler.step()
    path = str(tmp_path / "checkpoint.pth")
    save_checkpoint(a, oa, 1, path)
    b, ob = make(2)
    assert not torch.equal(a.prototypes, b.prototypes)
    assert load_checkpoint(b, ob, path) == 1
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert ob.global_step == 3 and ob.lr_scheduler.last_epoch == oa.lr_scheduler.last_epoch
    sa, sb = oa.optimizer.state_dict()["state"], ob.optimizer.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) for k in sa)
    assert load_checkpoint(b, ob, str(tmp_path / "missing.pth")) == 0


def test_backbone_loads_dino_style_checkpoints(tmp_path):
    """--model_path: a plain state_dict, and a DINO training checkpoint ({"teacher": {"module.backbone.<name>": ...}, "args":
    Namespace}) both fill the backbone."""
    import argparse

    import torch

    from timetuning_amd import synth
    from timetuning_amd.models import get_backbone

    cfg = synth.ARCHS["tiny-s16"]
    w = {k: torch.from_numpy(v) for k, v in synth.make_vit_weights(mode="stress", seed=7, **cfg).items()}
    p1, p2 = str(tmp_path / "plain.pth"), str(tmp_path / "dino_full.pth")
    torch.save(w, p1)
    torch.save({"teacher": {"module.backbone." + k: v for k, v in w.items()}, "args": argparse.Namespace(arch="vit_small"), "epoch": 3}, p2)
    for path in (p1, p2):
        m = get_backbone("dino-s16", path, vit_cfg=cfg)
        assert torch.equal(m.blocks[5].mlp.fc1.weight, w["blocks.5.mlp.fc1.weight"]) and torch.equal(m.pos_embed, w["pos_embed"])


def test_missing_model_path_raises_instead_of_training_from_random_weights():
    """only an EMPTY model path opts into synthetic weights; the CLI default / a typo must not silently do so."""
    from timetuning_amd import synth
    from timetuning_amd.models import FeatureExtractor, get_backbone

    cfg = synth.ARCHS["tiny-s16"]
    with pytest.raises(FileNotFoundError):
        get_backbone("dino-s16", "vits16_800ep.pth.tar", vit_cfg=cfg)
    with pytest.raises(FileNotFoundError):
        FeatureExtractor("dino-s16", "/nonexistent/typo.pth", [32, 16], vit_cfg=cfg)
    assert get_backbone("dino-s16", "", vit_cfg=cfg) is not None


def test_ddp_wrapper_checkpoint_keys_match_the_reference(tmp_path):
    """The reference wraps nn.parallel.DistributedDataParallel (models.py:1292-1295), so its multi-GPU checkpoints carry
    ``model.module.<name>`` keys: the wrapper here writes the same keys and loads such a checkpoint strictly."""
    import torch

    from timetuning_amd import synth
    from timetuning_amd.models import DistributedDataParallelModel, FeatureExtractor
    from timetuning_amd.time_tuning import TimeT

    def make(seed):
        fe = FeatureExtractor("dino-s16", "", [64, 32], unfreeze_layers=["blocks.11"], vit_cfg=synth.ARCHS["tiny-s16"], init="stress", seed=seed)
        m = TimeT(fe, 12, prototype_init=torch.from_numpy(synth.make_prototypes(12, 32, seed=seed)))
        m.init_momentum_teacher()
        return m

    inner = make(1)
    ddp = DistributedDataParallelModel(inner, 0)
    keys = list(ddp.state_dict().keys())
    assert keys and all(k.startswith("model.module.") for k in keys)
    assert {k[len("model.module."):] for k in keys} == set(inner.state_dict().keys())
    assert "model.module.teacher.backbone.blocks.3.attn.qkv.weight" in keys and "model.module.teacher_prototypes" in keys
    # a reference-style DDP checkpoint: {"model": {"model.module.<name>": tensor}}
    ref_style = {"model.module." + k: v.clone() + 1.0 for k, v in make(2).state_dict().items()}
    path = str(tmp_path / "ddp.pth")
    torch.save({"model": ref_style}, path)
    ddp.load_state_dict(torch.load(path)["model"])                      # strict
    assert torch.equal(inner.prototypes.detach(), ref_style["model.module.prototypes"])
    assert ddp.get_non_ddp_model() is inner and ddp.prototypes is inner.prototypes


def test_kmeans_empty_cluster_split_terminates():
    """faiss' donor loop has no accepting donor when n == k or every cluster holds one point; it must not spin."""
    import numpy as np
    import torch

    from timetuning_amd.clustering import Kmeans

    km = Kmeans.__new__(Kmeans)
    cent = torch.arange(12.0).view(4, 3).clone()
    before = cent.clone()
    counts = np.array([1, 0, 1, 1])
    assert km._split_empty(cent, counts, 4) == 0 and torch.equal(cent, before)          # n == k: nothing to split
    counts = np.array([1, 0, 1, 1])
    assert km._split_empty(cent, counts, 9) == 0                                        # all singletons
    counts = np.array([5, 0, 1, 1])
    assert km._split_empty(cent, counts, 7) == 1 and counts.tolist() == [3, 2, 1, 1]    # the populated donor is split


def test_step_graph_refuses_the_runtimes_packet_capture(monkeypatch):
    """Round 6: ROCm 7.2's packet-captured hipGraph replay corrupts TimeT's captured step.  The package switches it off
    on import; where somebody asked for it to stay on - or the package was imported after the GPU had been initialised -
    ``TimeT.enable_step_graph`` refuses instead of replaying a graph nobody can trust.  Host logic only: no GPU call."""
    import timetuning_amd
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.time_tuning import TimeT
    from timetuning_amd import synth

    assert os.environ.get(timetuning_amd.GRAPH_FLAG) == "0" and timetuning_amd.step_graph_safe()
    assert os.environ.get("GPU_MAX_HW_QUEUES") is not None
    cfg = synth.ARCHS["tiny-s16"]
    fe = FeatureExtractor("dino-s16", "", [32, 32, 16, 8], unfreeze_layers=["blocks.11"], vit_cfg=cfg, init="stress")
    m = TimeT(fe, 5)
    m.enable_step_graph()                       # fine
    m.enable_step_graph(False)
    monkeypatch.setenv(timetuning_amd.GRAPH_FLAG, "1")
    assert not timetuning_amd.step_graph_safe()
    with pytest.raises(RuntimeError, match="packet capture"):
        m.enable_step_graph()
    monkeypatch.setenv(timetuning_amd.GRAPH_FLAG, "0")
    monkeypatch.setattr(timetuning_amd, "_LATE", True)   # imported after the runtime came up
    with pytest.raises(RuntimeError, match="packet capture"):
        m.enable_step_graph()
