#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

It imports the reference's own Python (``time_tuning.TimeT``, ``models.FeatureExtractor``,
``my_utils.sinkhorn``, ``mask_propagation.propagate_labels``, ``time_tuning.SwavOptimizer``)
with empty stand-in modules for third-party imports the image lacks (they are never executed
on this path), drives it on seeded inputs built by ``timetuning_amd.synth`` and writes the
input/output tensors to ``tests/golden/*.npz``.  Nothing from the reference is copied; the
fixtures are data.  Recipe: SURVEY.md Appendix A.

    python oracle/gen_golden.py [--only NAME] [--full]
"""
from __future__ import annotations

import argparse
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

STUB_ROOTS = {"torchvision", "timm", "faiss", "cv2", "skimage", "tensorboard", "wandb", "nbformat", "mmcv",
              "pytorch_lightning", "torchmetrics", "matplotlib", "sklearn", "PIL", "tqdm", "seaborn", "imageio",
              "joblib", "kornia", "albumentations"}
STUB_EXACT = {"scipy.misc", "torch.utils.tensorboard"}


class _StubModule(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if fullname in STUB_EXACT or (root in STUB_ROOTS and not _really_importable(root)):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


_REAL = {}


def _really_importable(root):
    if root not in _REAL:
        _REAL[root] = any(os.path.isdir(os.path.join(p, root)) or os.path.isfile(os.path.join(p, root + ".py"))
                          for p in sys.path if p and p != REF)
    # tqdm/sklearn/joblib exist in this image; use the real ones
    return _REAL[root]


def import_reference():
    import torch

    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import anyio

    if not hasattr(anyio, "maybe_async"):
        anyio.maybe_async = None
    import dino_vision_transformer as dvt

    state = {"cfg": None}

    def fake_hub_load(repo, name, *a, **k):
        cfg = state["cfg"]
        from functools import partial

        return dvt.VisionTransformer(patch_size=cfg["patch_size"], embed_dim=cfg["embed_dim"], depth=cfg["depth"],
                                     num_heads=cfg["num_heads"], mlp_ratio=4, qkv_bias=True,
                                     norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))

    torch.hub.load = fake_hub_load
    import time_tuning as tt
    import my_utils
    import mask_propagation as mp
    import models

    class _W:
        def add_scalar(self, *a, **k):
            pass

    tt.writer = _W()

    # --use_mask branch: models.process_attentions calls torchvision's GaussianBlur and skimage's label, neither of
    # which is installed here.  The oracle's restatements of their published algorithms stand in for them (and are
    # declared "parity unpinned" there); the reference's own code around them then runs unchanged.
    import timet_oracle as orc

    class _GaussianBlurStandIn:
        def __init__(self, kernel_size, sigma):
            self.kernel_size, self.sigma = kernel_size, sigma

        def __call__(self, img):
            return orc.gaussian_blur(img, self.kernel_size, self.sigma)

    models.GaussianBlur = _GaussianBlurStandIn
    models.label = orc.label_components
    return dict(tt=tt, my_utils=my_utils, mp=mp, models=models, state=state, writer=_W())


def t2n(t):
    return t.detach().cpu().numpy().copy()  # copy: state_dict tensors alias live parameters


def build_reference_model(ref, arch, cfg, K, head_list, mode, seed, teacher=False, queue=0):
    import torch

    from timetuning_amd import synth

    ref["state"]["cfg"] = cfg
    tt = ref["tt"]
    fe = ref["models"].FeatureExtractor(arch, "", list(head_list), unfreeze_layers=["blocks.11", "blocks.10"])
    bb = {k: torch.from_numpy(v) for k, v in synth.make_vit_weights(mode=mode, seed=seed, **cfg).items()}
    fe.backbone.load_state_dict(bb, strict=True)
    hd = {k: torch.from_numpy(v) for k, v in synth.make_head_weights(cfg["embed_dim"], head_list, mode=mode, seed=seed).items()}
    fe.head.load_state_dict(hd, strict=True)
    model = tt.TimeT(fe, K)
    with torch.no_grad():
        model.prototypes.copy_(torch.from_numpy(synth.make_prototypes(K, fe.feature_dim, seed)))
    if teacher:
        model.init_momentum_teacher()
    if queue:
        model.init_queue(queue)
    ref["mp"].mask_neighborhood = None
    return model


# ------------------------------------------------------------------------------------------


def gen_sinkhorn(ref):
    import torch

    from timetuning_amd import synth

    sk = ref["my_utils"].sinkhorn
    out = {}
    kat = torch.tensor([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]])
    out["kat_in"] = t2n(kat)
    out["kat_it3"] = t2n(sk(kat, 3))
    out["kat_it0"] = t2n(sk(kat, 0))
    for tag, (K, B, iters) in dict(a=(50, 392, 10), b=(200, 784, 10), c=(16, 40, 3), d=(7, 333, 1)).items():
        x = synth.normal(f"sk.x.{tag}", (B, 32))
        p = synth.normal(f"sk.p.{tag}", (K, 32))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        scores = torch.from_numpy(x @ p.T)
        q = sk(torch.exp(scores / 0.05).t(), iters)
        out[f"{tag}_scores"] = t2n(scores)
        out[f"{tag}_iters"] = np.int64(iters)
        out[f"{tag}_q"] = t2n(q)
    np.savez_compressed(os.path.join(OUT, "sinkhorn.npz"), **out)
    print("sinkhorn.npz", {k: v.shape for k, v in out.items()})


def _sk_worker(rank, W, port, scores_np, iters, ret):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, REF)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    ref = import_reference()
    B = scores_np.shape[0] // W
    sc = torch.from_numpy(scores_np[rank * B:(rank + 1) * B])
    q = ref["my_utils"].sinkhorn(torch.exp(sc / 0.05).t(), iters, W)
    ret[rank] = t2n(q)
    dist.destroy_process_group()


def gen_sinkhorn_dist(ref):
    """W=2 gloo run of the reference's distributed Sinkhorn (my_utils.py:250-272)."""
    import torch.multiprocessing as mp

    from timetuning_amd import synth

    K, Bg, iters, W = 50, 392, 10, 2
    x = synth.normal("skd.x", (Bg, 32))
    p = synth.normal("skd.p", (K, 32))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    scores = (x @ p.T).astype(np.float32)
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_sk_worker, args=(r, W, 29611, scores, iters, ret)) for r in range(W)]
    [p_.start() for p_ in procs]
    [p_.join() for p_ in procs]
    q = np.concatenate([ret[r] for r in range(W)], axis=0)
    np.savez_compressed(os.path.join(OUT, "sinkhorn_w2.npz"), scores=scores, q=q, iters=np.int64(iters), world_size=np.int64(W))
    print("sinkhorn_w2.npz", q.shape)


def gen_sinkhorn_w8(ref):
    """8 gloo ranks of the reference's distributed Sinkhorn (my_utils.py:250-272) at C3's shape; every 52nd row of each rank's q is
    kept (160 rows x 8 ranks), the scores are regenerated by ``synth.make_sinkhorn_w8_scores``."""
    import torch.multiprocessing as mp

    K, Bl, iters, W, stride = 200, 8320, 10, 8, 52
    from timetuning_amd import synth

    scores = synth.make_sinkhorn_w8_scores()
    assert scores.shape == (W * Bl, K) and np.all(scores * 1024 == np.rint(scores * 1024))
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_sk_worker, args=(r, W, 29617, scores, iters, ret)) for r in range(W)]
    [p_.start() for p_ in procs]
    [p_.join() for p_ in procs]
    q = np.stack([ret[r][::stride] for r in range(W)])        # [W, 160, K]
    rowsum = np.stack([ret[r].sum(1) for r in range(W)])      # [W, Bl]: every row of q sums to 1
    colsum = np.sum([ret[r].astype(np.float64).sum(0) for r in range(W)], axis=0)   # [K]: prototype usage over the global batch
    np.savez_compressed(os.path.join(OUT, "sinkhorn_w8.npz"), q=q, stride=np.int64(stride), iters=np.int64(iters), world_size=np.int64(W),
                        rows_per_rank=np.int64(Bl), rowsum_min=np.float64(rowsum.min()), rowsum_max=np.float64(rowsum.max()), colsum=colsum,
                        scores_checksum=np.float64(scores.astype(np.float64).sum()))
    print("sinkhorn_w8.npz", q.shape)


def gen_label_prop(ref):
    import torch

    from timetuning_amd import synth

    mp = ref["mp"]
    out = {}
    # window-mask counts
    out["mask_nnz_g14_r6"] = np.int64(mp.restrict_neighborhood(14, 14, 6).sum().item())
    out["mask_nnz_g28_r6"] = np.int64(mp.restrict_neighborhood(28, 28, 6).sum().item())

    class _M:  # label_propagation reads model.spatial_resolution for non-TimeT models (:404-405)
        def __init__(self, g):
            self.spatial_resolution = g

    cases = dict(a=(14, 4, 64, 50, 7, 6, 5), b=(14, 10, 32, 20, 7, 6, 5), c=(3, 2, 8, 3, 7, 1, 2), d=(14, 3, 48, 200, 1, 6, 5),
                 e=(28, 3, 16, 10, 7, 6, 5))
    for tag, (g, fs, D, K, nlast, r, topk) in cases.items():
        n = g * g
        base = synth.normal(f"lp.base.{tag}", (n, D))
        frames = [base]
        for t in range(1, fs):
            frames.append(np.roll(base.reshape(g, g, D), (t % g, (2 * t) % g), axis=(0, 1)).reshape(n, D)
                          + 0.3 * synth.normal(f"lp.noise.{tag}.{t}", (n, D)))
        feats = torch.from_numpy(np.stack(frames).astype(np.float32))
        q0 = np.abs(synth.normal(f"lp.q.{tag}", (n, K))).astype(np.float32)
        q0 /= q0.sum(1, keepdims=True)
        q0 = torch.from_numpy(q0)
        seed = q0.view(g, g, K).permute(2, 0, 1).unsqueeze(0)
        mp.mask_neighborhood = None
        maps = mp.propagate_labels(nlast, r, topk, _M(g), feats, seed, features_exist=True)
        out[f"{tag}_cfg"] = np.array([g, fs, D, K, nlast, r, topk], np.int64)
        out[f"{tag}_feats"] = t2n(feats)
        out[f"{tag}_q0"] = t2n(q0)
        out[f"{tag}_maps"] = t2n(torch.stack(maps))
    mp.mask_neighborhood = None
    np.savez_compressed(os.path.join(OUT, "label_prop.npz"), **out)
    print("label_prop.npz", {k: v.shape for k, v in out.items()})


def _grad_dict(model, names):
    g = {}
    for n, p in model.named_parameters():
        if p.grad is not None:
            g[n] = p.grad
    return {n: t2n(g[n]) for n in names}, {n: float(v.double().norm()) for n, v in g.items()}


def gen_timet_masked(ref, tag, *args):
    """--use_mask fixtures: smooth clips, and a search over clip seeds for a run in which the reference does not hit
    its IndexError on small mask components (models.py:127-130) in any frame of any step."""
    for seed0 in range(0, 400, 10):
        try:
            gen_timet(ref, tag, *args, use_mask=True, clip_seed0=seed0)
            print(f"  {tag}: reference survived with clip seed base {seed0}")
            return
        except IndexError as e:
            if "mask" not in str(e):
                raise
            print(f"  {tag}: clip seed base {seed0} trips the reference's small-component IndexError, trying the next")
    raise RuntimeError("no surviving seed found")


def gen_timet(ref, tag, arch, cfg, K, head_list, bs, fs, mode, teacher, queue, steps, full_tensors, use_mask=False, clip_seed0=0):
    """One or more reference training iterations; saves loss, labels, grads, post-step parameters."""
    import torch

    from timetuning_amd import synth

    tt = ref["tt"]
    model = build_reference_model(ref, arch, cfg, K, head_list, mode, 1, teacher=teacher, queue=queue)
    model.train()
    E, I = 1, max(steps + 1, 4)
    swav = tt.SwavOptimizer(model, "AdamW", True, 1e-4 / 10, 1e-4, "CosineAnnealingLR",
                            ref["my_utils"].cosine_scheduler(0.04, 0.4, E, I), I, E)
    assert swav.lr_scheduler is not None
    if teacher:
        model.set_momentum_teacher_schedular_params(0.995, 1.0, E, I)
    out = dict(cfg=np.array([bs, fs, K, int(teacher), queue, steps, E, I], np.int64),
               head_list=np.array(head_list, np.int64))
    out["arch"] = np.array(arch)
    out["mode"] = np.array(mode)
    out["use_mask"] = np.int64(use_mask)
    out["clip_seed0"] = np.int64(clip_seed0)
    out["vit_cfg"] = np.array([cfg["embed_dim"], cfg["depth"], cfg["num_heads"], cfg["patch_size"]], np.int64)
    watch = ["prototypes", "feature_extractor.head.6.weight", "feature_extractor.head.0.bias",
             "feature_extractor.backbone.blocks.11.mlp.fc2.weight", "feature_extractor.backbone.blocks.10.attn.qkv.weight",
             "feature_extractor.backbone.blocks.10.norm1.weight", "feature_extractor.backbone.blocks.11.attn.proj.bias"]
    n_tok = (224 // cfg["patch_size"]) ** 2
    for s in range(steps):
        if use_mask:
            x = torch.from_numpy(synth.make_smooth_clips(bs, fs, 224, seed=clip_seed0 + 1 + s))
        else:
            x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1 + s, coherent=True))
        torch.manual_seed(100 + s)
        perm = torch.randperm(bs * n_tok)
        torch.manual_seed(100 + s)
        if s == 0:
            with torch.no_grad():
                feats, attn = model.feature_extractor(x.view(bs * fs, 3, 224, 224))
                bfeats, _ = model.feature_extractor(x.view(bs * fs, 3, 224, 224), use_head=False)
            if full_tensors:
                out["features"] = t2n(feats)
                out["backbone_features"] = t2n(bfeats)
                out["attn_cls_row"] = t2n(attn[:, :, 0, :])
            else:
                out["features_slice"] = t2n(feats[:, ::49, ::16])
                out["backbone_features_slice"] = t2n(bfeats[:, ::49, ::16])
                out["features_norm"] = np.float64(feats.double().norm().item())
                out["backbone_features_norm"] = np.float64(bfeats.double().norm().item())
                out["attn_cls_row"] = t2n(attn[:, :, 0, ::7])
            torch.manual_seed(100 + s)
        out[f"perm{s}"] = t2n(perm)
        # replay get_loss's internals we want to pin (labels, q) without touching the model state:
        if use_mask and s == 0:  # the masks get_loss is about to build, for diagnosis of a mismatch
            with torch.no_grad():
                _, attn_s = model.feature_extractor(x.view(bs * fs, 3, 224, 224))
                out["student_mask0"] = t2n(ref["models"].process_attentions(attn_s, model.feature_extractor.spatial_resolution))
            torch.manual_seed(100 + s)
        loss = model(x, None, True, use_mask)
        out[f"loss{s}"] = np.float64(loss.item())
        swav.optimizer.zero_grad()
        loss.backward()
        gsel, gnorm = _grad_dict(model, watch)
        for n, v in gsel.items():
            out[f"grad{s}:{n}"] = v if (full_tensors or v.size < 70000) else v.reshape(-1)[::97].copy()
        out[f"gradnorm_names{s}"] = np.array(sorted(gnorm))
        out[f"gradnorm{s}"] = np.array([gnorm[k] for k in sorted(gnorm)], np.float64)
        # optimizer.step without the zero_grad/backward that SwavOptimizer.step would redo
        swav.optimizer.step()
        swav.lr_scheduler.step()
        swav.global_step += 1
        for pg in swav.optimizer.param_groups:
            if pg["weight_decay"] != 0:
                pg["weight_decay"] = swav.wd_schedule[swav.global_step]
        model.normalize_prototypes()
        if teacher:
            model.update_momentum_teacher(swav.global_step, ref["writer"])
        sd = model.state_dict()
        for n in watch:
            v = t2n(sd[n])
            out[f"param{s}:{n}"] = v if (full_tensors or v.size < 70000) else v.reshape(-1)[::97].copy()
        if teacher:
            out[f"teacher_prototypes{s}"] = t2n(sd["teacher_prototypes"])
            v = t2n(sd["teacher.backbone.blocks.11.mlp.fc2.weight"])
            out[f"teacher_fc2_{s}"] = v if full_tensors else v.reshape(-1)[::97].copy()
            v = t2n(sd["teacher.backbone.blocks.3.attn.qkv.weight"])
            out[f"teacher_b3qkv_{s}"] = v.reshape(-1)[::97].copy()
        if queue:
            out[f"queue_head{s}"] = t2n(model.queue[: min(64, queue)])
            out[f"queue_sum{s}"] = np.float64(model.queue.double().sum().item())
        out[f"lr{s}"] = np.array([pg["lr"] for pg in swav.optimizer.param_groups], np.float64)
        out[f"wd{s}"] = np.array([pg["weight_decay"] for pg in swav.optimizer.param_groups], np.float64)
        print(f"  {tag} step {s}: loss {loss.item():.6f}")
    out["state_dict_keys"] = np.array(list(model.state_dict().keys()))
    out["group_sizes"] = np.array([len(pg["params"]) for pg in swav.optimizer.param_groups], np.int64)
    np.savez_compressed(os.path.join(OUT, f"timet_{tag}.npz"), **out)
    print(f"timet_{tag}.npz written ({len(out)} arrays)")


def gen_aux(ref, tag, arch, cfg, K, head_list, bs, fs, mode):
    """Intermediate tensors of get_loss (q, target scores, labels, p_map) via the reference's own methods."""
    import torch

    from timetuning_amd import synth

    model = build_reference_model(ref, arch, cfg, K, head_list, mode, 1)
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1, coherent=True))
    with torch.no_grad():
        feats, _ = model.feature_extractor(x.view(bs * fs, 3, 224, 224))
        bfeats, _ = model.feature_extractor(x.view(bs * fs, 3, 224, 224), use_head=False)
        feats = feats.view(bs, fs, *feats.shape[1:])
        bfeats = bfeats.view(bs, fs, *bfeats.shape[1:])
        q, _ = model.get_scores(feats[:, 0], 0.05, 10)
        _, tscores = model.get_scores(feats[:, -1], 0.05, 10)
        labels, pmaps = [], []
        for i in range(bs):
            maps = model.make_seg_maps(q[i], bfeats[i], 7, 6, 5, features_exist=True)
            pmaps.append(maps[-1])
            labels.append(maps[-1].unsqueeze(0).argmax(dim=1)[0])
    srt = torch.sort(torch.stack(pmaps).flatten(2), dim=1, descending=True).values
    np.savez_compressed(os.path.join(OUT, f"aux_{tag}.npz"), q=t2n(q), target_scores=t2n(tscores),
                        labels=t2n(torch.stack(labels)), p_map=t2n(torch.stack(pmaps)),
                        label_margin=t2n(srt[:, 0] - srt[:, 1]),
                        cfg=np.array([bs, fs, K], np.int64), head_list=np.array(head_list, np.int64),
                        vit_cfg=np.array([cfg["embed_dim"], cfg["depth"], cfg["num_heads"], cfg["patch_size"]], np.int64))
    print(f"aux_{tag}.npz written")


def gen_davis_protocol(ref):
    """The evaluation protocol of mask_propagation.py:821-830 (n_last_frames 4, neighbourhood 12, top-5, bilinear
    upsampling to the input resolution, arg-max) driven through the reference's own propagate_labels / to_one_hot and
    torch ops, on synthetic token features that follow a drifting annotation."""
    import torch

    from timetuning_amd import synth

    mp = ref["mp"]

    class _M:
        def __init__(self, g):
            self.spatial_resolution = g

    out = {}
    for tag, (g, fs, D, C, R) in dict(a=(14, 8, 64, 3, 112), b=(28, 7, 32, 4, 112), c=(14, 3, 16, 2, 56)).items():
        n = g * g
        yy, xx = np.mgrid[0:R, 0:R]
        ann = np.zeros((R, R), np.int64)
        for o in range(1, C):  # C-1 discs on a background
            cy, cx, rad = R * (0.25 + 0.5 * (o - 1) / max(C - 2, 1)), R * (0.3 + 0.15 * o), R * 0.16
            ann[(yy - cy) ** 2 + (xx - cx) ** 2 < rad ** 2] = o
        proto = synth.normal(f"dv.proto.{tag}", (C, D))
        frames = []
        for t in range(fs):  # token features = prototype of the (shifted) label under the token centre + noise
            sh = np.roll(ann, (2 * t * R // 112, 3 * t * R // 112), axis=(0, 1))
            centres = ((np.arange(g) + 0.5) * R / g).astype(np.int64)
            lab = sh[np.ix_(centres, centres)].reshape(n)
            frames.append(proto[lab] + 0.6 * synth.normal(f"dv.noise.{tag}.{t}", (n, D)))
        feats = torch.from_numpy(np.stack(frames).astype(np.float32))
        first = torch.from_numpy(ann)
        mp.mask_neighborhood = None
        pred = mp.propagate_labels(4, 12, 5, _M(g), feats, mp.to_one_hot(first.unsqueeze(0)).unsqueeze(0), features_exist=True)
        maps = torch.stack(pred, dim=0)
        up = torch.nn.functional.interpolate(maps, size=(R, R), mode="bilinear", align_corners=False)
        _, labels = torch.max(up, dim=1)
        top2 = up.topk(2, dim=1).values
        out[f"{tag}_cfg"] = np.array([g, fs, D, C, R], np.int64)
        out[f"{tag}_feats"] = t2n(feats)
        out[f"{tag}_annotation"] = ann.astype(np.uint8)
        out[f"{tag}_maps"] = t2n(maps)
        out[f"{tag}_pred"] = t2n(labels).astype(np.uint8)
        out[f"{tag}_near_tie"] = t2n((top2[:, 0] - top2[:, 1]) < 1e-9)
        print(f"  {tag}: label histogram of the last frame {np.bincount(t2n(labels)[-1].reshape(-1), minlength=C)}")
    mp.mask_neighborhood = None
    np.savez_compressed(os.path.join(OUT, "davis_protocol.npz"), **out)
    print("davis_protocol.npz written")


def gen_extractor_sizes(ref):
    """FeatureExtractor on inputs that are NOT 224x224: interpolate_pos_encoding's bicubic branch
    (dino_vision_transformer.py:219-234), non-square token grids, token counts other than 197."""
    import torch

    from timetuning_amd import synth

    cfg = synth.ARCHS["tiny-s16"]
    model = build_reference_model(ref, "dino-s16", cfg, 20, (128, 128, 64, 32), "stress", 1)
    fe = model.feature_extractor
    out = dict(vit_cfg=np.array([cfg["embed_dim"], cfg["depth"], cfg["num_heads"], cfg["patch_size"]], np.int64),
               head_list=np.array((128, 128, 64, 32), np.int64))
    for tag, (H, W) in dict(a=(160, 192), b=(256, 256), c=(96, 64)).items():
        x = torch.from_numpy(synth.normal(f"sizes.x.{tag}", (2, 3, H, W)))
        with torch.no_grad():
            feats, attn = fe(x)
            bfeats, _ = fe(x, use_head=False)
            pos = fe.backbone.interpolate_pos_encoding(torch.zeros(1, 1 + (H // 16) * (W // 16), cfg["embed_dim"]), H, W)
        out[f"{tag}_hw"] = np.array([H, W], np.int64)
        out[f"{tag}_pos"] = t2n(pos[0])
        out[f"{tag}_features"] = t2n(feats)
        out[f"{tag}_backbone_features"] = t2n(bfeats)
        out[f"{tag}_attn_cls_row"] = t2n(attn[:, :, 0, :])
    np.savez_compressed(os.path.join(OUT, "extractor_sizes.npz"), **out)
    print("extractor_sizes.npz", {k: v.shape for k, v in out.items()})


def gen_metric(ref):
    """metrics.PredsmIoU.compute (metrics.py:246-432): Hungarian / many-to-one / precision-based matching, with and without the
    background class, on label maps whose clusters are noisy refinements of the ground truth; and clustering.proto_clustering
    without the k-means merge (clustering.py:82-104; the merge needs faiss)."""
    import importlib

    import torch

    metrics = importlib.import_module("metrics")
    clustering = importlib.import_module("clustering")
    out = {}
    rng = np.random.default_rng(5)
    cases = dict(a=(6, 6, 4000), b=(4, 9, 6000), c=(3, 3, 500), d=(5, 2, 3000))
    for tag, (n_gt, n_pred, n) in cases.items():
        gt = rng.integers(0, n_gt, n)
        if tag == "b":
            gt = gt * 3                                   # non-contiguous label values
        pred = (gt // (3 if tag == "b" else 1)) % n_pred
        split = rng.random(n) < 0.5                        # over-segmentation: half of every class moves to another cluster id
        pred = np.where(split, (pred + n_gt) % n_pred, pred)
        noise = rng.random(n) < 0.15
        pred = np.where(noise, rng.integers(0, n_pred, n), pred) + (7 if tag == "c" else 0)
        out[f"{tag}_gt"], out[f"{tag}_pred"] = gt.astype(np.int16), pred.astype(np.int16)
        for involve_bg in (False, True):
            for mode, kw in dict(hungarian=dict(), many=dict(many_to_one=True), many_prec=dict(many_to_one=True, precision_based=True)).items():
                m = metrics.PredsmIoU(n_pred, n_gt, involve_bg=involve_bg)
                m.n_jobs = 1
                m.update(torch.from_numpy(gt), torch.from_numpy(pred))
                score, tp, fp, fn, reordered, bg = m.compute(True, **kw)
                key = f"{tag}_{mode}_{int(involve_bg)}"
                out[key + "_score"] = np.float64(score)
                ks = sorted(tp)
                out[key + "_classes"] = np.array(ks, np.int64)
                out[key + "_tp"] = np.array([tp[k] for k in ks], np.int64)
                out[key + "_fp"] = np.array([fp[k] for k in ks], np.int64)
                out[key + "_fn"] = np.array([fn[k] for k in ks], np.int64)
                out[key + "_reordered"] = np.asarray(reordered).astype(np.int16)
                out[key + "_bg"] = np.float64(bg)
    # proto_clustering (no merge)
    from timetuning_amd import synth
    x = torch.from_numpy(synth.normal("pc.x", (3, 196, 64)))
    protos = torch.from_numpy(synth.normal("pc.p", (12, 64)))
    assign = clustering.proto_clustering(x, protos, input_size=14, output_size=56)
    with torch.no_grad():
        xn, pn = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(protos, dim=-1)
        sc = torch.einsum("klm,nm->kln", xn, pn).permute(0, 2, 1).reshape(3, 12, 14, 14)
        up = torch.nn.functional.interpolate(sc, size=(56, 56), mode="bilinear", align_corners=False)
        top2 = up.topk(2, dim=1).values
    out["pc_assign"] = t2n(assign).astype(np.int16)
    out["pc_near_tie"] = t2n((top2[:, 0] - top2[:, 1]) < 1e-6)
    np.savez_compressed(os.path.join(OUT, "evaluator.npz"), **out)
    print("evaluator.npz written", len(out), "arrays")


def gen_transforms(ref):
    """The reference's training transforms (time_tuning.py:588-593) run on PIL clips with seeded generators.
    video_transformations.py reaches torchvision for ToTensor and the adjust_* functions; torchvision is not installed, so
    the published PIL-backend implementations of those five functions (they only call Pillow, which IS installed) stand in."""
    import importlib
    import random
    import types

    import torch
    from PIL import Image, ImageEnhance

    vt = importlib.import_module("video_transformations")

    def to_tensor(pic):
        a = np.array(pic, copy=True)
        return torch.from_numpy(a).view(pic.size[1], pic.size[0], 3).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    def adjust_hue(img, hue_factor):
        h, s_, v = img.convert("HSV").split()
        nh = (np.array(h, dtype=np.int32) + int(hue_factor * 255) % 256) % 256    # np_h += np.uint8(hue_factor * 255), wrapping
        return Image.merge("HSV", (Image.fromarray(nh.astype(np.uint8), "L"), s_, v)).convert("RGB")

    functional = types.SimpleNamespace(adjust_brightness=lambda img, f: ImageEnhance.Brightness(img).enhance(f),
                                       adjust_contrast=lambda img, f: ImageEnhance.Contrast(img).enhance(f),
                                       adjust_saturation=lambda img, f: ImageEnhance.Color(img).enhance(f), adjust_hue=adjust_hue)
    vt.torchvision = types.SimpleNamespace(transforms=types.SimpleNamespace(ToTensor=lambda: to_tensor, functional=functional))

    R = 64
    rand_color_jitter = vt.RandomApply([vt.ColorJitter(brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2)], p=0.8)
    data_transform = vt.Compose([rand_color_jitter, vt.RandomGrayscale(), vt.RandomGaussianBlur()])
    video_transform = vt.Compose([vt.Resize(R), vt.RandomResizedCrop((R, R)), vt.RandomHorizontalFlip(),
                                  vt.ClipToTensor(mean=[0.485, 0.456, 0.406], std=[0.228, 0.224, 0.225])])
    rng = np.random.default_rng(3)
    out = {}
    for tag, (H, W) in dict(a=(96, 128), b=(150, 100)).items():
        yy, xx = np.mgrid[0:H, 0:W]
        frames = []
        for t in range(3):
            base = np.stack([127 + 100 * np.sin((xx + 5 * t) / 9.0 + c) * np.cos((yy - 3 * t) / 7.0 + 2 * c) for c in range(3)], -1)
            frames.append(np.clip(base + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8))
        frames = np.stack(frames)
        out[f"{tag}_frames"] = frames
        for seed in range(7 if tag == "a" else 3):
            random.seed(seed)
            torch.manual_seed(seed)
            clip = [Image.fromarray(f) for f in frames]
            clip = data_transform(clip)
            if tag == "a":
                out[f"{tag}_seed{seed}_after_frame_transform"] = np.stack([np.array(im) for im in clip])
            out[f"{tag}_seed{seed}"] = t2n(video_transform(clip))
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **out)
    print("transforms.npz written", len(out), "arrays")


def gen_mask(ref):
    """models.process_attentions (with the blur / component-labelling stand-ins) on synthetic attention maps:
    peaked random maps at g = 14 and 28, plus hand-made cases for the small-component rule."""
    import torch

    out = {}
    for g, F_, H in ((14, 8, 6), (28, 6, 6), (7, 6, 2)):
        n = g * g
        gen = torch.Generator().manual_seed(7 + g)
        # speckle amplitude grows with the frame index: smooth maps (no small components, the reference survives) first
        amp = torch.tensor([0.02, 0.05, 0.1, 0.2, 0.4, 0.8, 1.5, 2.5])[:F_].view(F_, 1, 1, 1)
        logits = amp * torch.randn(F_, H, n + 1, n + 1, generator=gen)
        yy, xx = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
        for f in range(F_):  # a blob per frame so that the kept mass is spatially coherent, plus speckle
            cy, cx = float(2 + 3 * f % g), float(g - 3 - 2 * f % g)
            blob = 3.0 * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2.0 * (g / 5.0) ** 2))
            logits[f, :, 0, 1:] += blob.reshape(-1)
        attn = logits.softmax(dim=-1)
        # The reference indexes a [g,g] tensor with the [1,g,g] component mask (models.py:127-130), which raises
        # IndexError as soon as a frame HAS a component of <= 2 pixels.  Frames are therefore run one at a time: where
        # the reference survives its mask is the expected value (ref_ok = 1); where it raises, the fixture only records
        # that fact and the intended behaviour (drop the small components) is checked oracle-vs-HIP in the tests.
        masks, ok = [], []
        for f in range(F_):
            try:
                masks.append(t2n(ref["models"].process_attentions(attn[f:f + 1], g))[0])
                ok.append(1)
            except IndexError:
                masks.append(np.zeros((1, g, g), np.float32))
                ok.append(0)
        out[f"attn_cls_g{g}"] = t2n(attn[:, :, 0, :])
        out[f"mask_g{g}"] = np.stack(masks)
        out[f"ref_ok_g{g}"] = np.array(ok, np.int64)
        print(f"  g={g}: reference survived {sum(ok)}/{F_} frames")
    np.savez_compressed(os.path.join(OUT, "attention_mask.npz"), **out)
    print("attention_mask.npz written")


def gen_scaler(ref):
    """my_utils.normalize_and_transform (my_utils.py:19-37) with the REAL scikit-learn StandardScaler (installed here) and
    faiss.PCAMatrix (not installed) replaced by a recorder: what it is trained on IS the reference's standardised features."""
    import torch

    from timetuning_amd import synth

    mu = ref["my_utils"]
    seen = {}

    class _RecordingPCA:
        def __init__(self, d, k):
            self.k, self.is_trained = k, True

        def train(self, feats):
            seen["z"] = np.array(feats, copy=True)

        def apply_py(self, feats):
            return feats[:, : self.k]

    mu.faiss.PCAMatrix = _RecordingPCA
    mu.normalize_and_transform(torch.from_numpy(synth.make_scaler_features()), 3)
    z = seen["z"]
    rows = np.r_[0:32, 99984:100016, 199984:200016, z.shape[0] - 32:z.shape[0]]
    np.savez_compressed(os.path.join(OUT, "scaler.npz"), rows=rows, z_rows=z[rows], z_colsum=z.astype(np.float64).sum(0),
                        z_colsumsq=(z.astype(np.float64) ** 2).sum(0), shape=np.array(z.shape))
    print("scaler.npz", z.shape, z.dtype, z[rows].std(0))


def gen_sched(ref):
    cs = ref["my_utils"].cosine_scheduler
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), wd_1_4=cs(0.04, 0.4, 1, 4), ema_2_5=cs(0.995, 1.0, 2, 5),
                        wd_3_7=cs(0.04, 0.4, 3, 7))
    print("schedules.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--full", action="store_true", help="also the full-size ViT-S/16 C1 fixture (minutes of CPU)")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    from timetuning_amd import synth

    tiny = synth.ARCHS["tiny-s16"]
    jobs = {
        "sched": lambda: gen_sched(ref),
        "scaler": lambda: gen_scaler(ref),
        "sinkhorn": lambda: gen_sinkhorn(ref),
        "sinkhorn_w2": lambda: gen_sinkhorn_dist(ref),
        "sinkhorn_w8": lambda: gen_sinkhorn_w8(ref),
        "label_prop": lambda: gen_label_prop(ref),
        "aux_tiny": lambda: gen_aux(ref, "tiny", "dino-s16", tiny, 20, (128, 128, 64, 32), 2, 3, "stress"),
        "timet_tiny": lambda: gen_timet(ref, "tiny", "dino-s16", tiny, 20, (128, 128, 64, 32), 2, 3, "stress", False, 0, 3, True),
        "timet_tiny_tq": lambda: gen_timet(ref, "tiny_tq", "dino-s16", tiny, 20, (128, 128, 64, 32), 2, 2, "stress", True, 40, 3, True),
        # 300 prototypes (BASELINE config C4 has 400): the K > 256 instances of the propagation / Sinkhorn / cross-entropy kernels
        "timet_tiny_k300": lambda: gen_timet(ref, "tiny_k300", "dino-s16", tiny, 300, (128, 128, 64, 32), 2, 3, "stress", False, 0, 2, True),
        # six-frame clips (BASELINE config C4 has eight): up to five context frames per target in the label propagation
        "timet_tiny_f6": lambda: gen_timet(ref, "tiny_f6", "dino-s16", tiny, 20, (128, 128, 64, 32), 1, 6, "stress", False, 0, 2, True),
        # patch size 8 (BASELINE config C5's shape: 28 x 28 token grid, 785 tokens, the KV-tiled attention and the general patch-embed
        # kernel on the GPU side), on a narrow ViT so that the reference's CPU run stays short
        "timet_tiny_s8": lambda: gen_timet(ref, "tiny_s8", "dino-s8", dict(embed_dim=64, depth=12, num_heads=1, patch_size=8), 12,
                                           (64, 64, 32, 16), 1, 3, "stress", False, 0, 2, True),
        "mask": lambda: gen_mask(ref),
        "transforms": lambda: gen_transforms(ref),
        "metric": lambda: gen_metric(ref),
        "extractor_sizes": lambda: gen_extractor_sizes(ref),
        "davis_protocol": lambda: gen_davis_protocol(ref),
        "timet_tiny_mask": lambda: gen_timet_masked(ref, "tiny_mask", "dino-s16", tiny, 20, (128, 128, 64, 32), 2, 3, "dino", False, 0, 2, True),
        "timet_tiny_mask_tq": lambda: gen_timet_masked(ref, "tiny_mask_tq", "dino-s16", tiny, 20, (128, 128, 64, 32), 2, 2, "dino", True, 40, 2,
                                                       True),
    }
    if a.full:
        s16 = synth.ARCHS["dino-s16"]
        jobs["timet_c1"] = lambda: gen_timet(ref, "c1", "dino-s16", s16, 50, (1024, 1024, 512, 256), 2, 2, "stress", False, 0, 1, False)
    for name, fn in jobs.items():
        if a.only and a.only != name:
            continue
        print("==", name)
        fn()


if __name__ == "__main__":
    main()
