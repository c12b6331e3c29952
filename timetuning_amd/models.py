"""FeatureExtractor and the data-parallel wrapper - the reference's ``models.py`` surface for the
training path (``models.py:903-1078``, ``:1292-1306``), backed by the HIP kernels.

Only the ``"dino-*"`` branch of ``get_features`` (``models.py:965-969``) exists: BASELINE's configs
use nothing else, and the other backbones are out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import dino_vision_transformer as dvt
from . import engine, synth
from . import hip_ops as ops

# models.py:76
spatial_resolutions = {"dino-s16": 14, "dino-b16": 14, "dino-s8": 28}


def get_backbone(name: str, model_path: str = "", vit_cfg: Optional[dict] = None, init: str = "dino", seed: int = 1):
    """models.py:773-900 for the DINO entries.  The reference downloads pretrained weights through
    ``torch.hub`` (:780-785); there is no network here, so the same architecture is built locally and
    filled from ``model_path`` (a state_dict or a checkpoint with a ``"model"``/``"state_dict"`` entry).  ONLY an empty
    ``model_path`` selects the portable synthetic generator (non-pretrained weights: benchmarks and parity tests); a
    non-empty path that does not exist raises ``FileNotFoundError`` instead of silently training from random weights.
    Unknown names raise (the reference prints the error and then fails on ``None.eval()``, :897-900)."""
    if vit_cfg is None:
        if name not in synth.ARCHS or name not in spatial_resolutions:
            raise ValueError(f"unknown architecture {name!r}; built: {sorted(spatial_resolutions)}")
        vit_cfg = synth.ARCHS[name]
    cfg = dict(vit_cfg)
    model = dvt.VisionTransformer(patch_size=cfg["patch_size"], embed_dim=cfg["embed_dim"], depth=cfg["depth"],
                                  num_heads=cfg["num_heads"], mlp_ratio=4, qkv_bias=True)
    import os

    if model_path and not os.path.isfile(model_path):
        raise FileNotFoundError(f"--model_path {model_path!r} does not exist.  The reference downloads the pretrained DINO weights "
                                "through torch.hub (models.py:780-785); there is no network here, so pass a local DINO checkpoint, "
                                "or pass an EMPTY model path (--model_path \"\") to opt into synthetic, non-pretrained weights.")
    if model_path:
        # full DINO training checkpoints carry an argparse.Namespace next to the "teacher" / "student" dicts
        sd = torch.load(model_path, map_location="cpu", weights_only=False)
        for key in ("model", "state_dict", "teacher", "student"):
            if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
                sd = sd[key]
                break
        sd = {k.replace("module.", "").replace("backbone.", ""): v for k, v in sd.items()}
        missing = model.load_state_dict(sd, strict=False)
        if missing.missing_keys:
            raise RuntimeError(f"{model_path}: missing backbone tensors {missing.missing_keys[:4]} ...")
    else:
        weights = synth.make_vit_weights(mode=init, seed=seed, **cfg)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    model.eval()
    return model


class FeatureExtractor(nn.Module):
    """``FeatureExtractor(arcitecture, model_path, head_layer_list=[], unfreeze_layers=[], kqv="all")``
    (``models.py:903-935``, argument spelling kept).  ``forward(x, use_head=True) -> (features, attentions)``.

    Extra keyword-only arguments (not in the reference) select the synthetic initialisation used when no
    checkpoint is available, and ``return_attention=False`` skips the second output, which the reference pays a
    second full backbone pass for (:968) although nothing on the default training path reads it."""

    def __init__(self, arcitecture, model_path="", head_layer_list: Sequence[int] = (), unfreeze_layers: Sequence[str] = (),
                 kqv="all", *, vit_cfg: Optional[dict] = None, init: str = "dino", seed: int = 1, return_attention: bool = True):
        super().__init__()
        self.backbone = get_backbone(arcitecture, model_path, vit_cfg, init, seed)
        self.freeze_backbone(unfreeze_layers=unfreeze_layers)
        self.architecture = arcitecture
        self.kqv = kqv
        self.return_attention = return_attention
        self.feature_dim = self.backbone.embed_dim  # the reference probes this with a 224x224 forward (:911-912)
        self.spatial_resolution = spatial_resolutions.get(arcitecture, 224 // self.backbone.patch_embed.patch_size)
        self.head = None
        head_layer_list = list(head_layer_list)
        if len(head_layer_list):
            layers: List[nn.Module] = [nn.Linear(self.feature_dim, head_layer_list[0]), nn.GELU()]
            for i in range(1, len(head_layer_list)):
                layers.append(nn.Linear(head_layer_list[i - 1], head_layer_list[i]))
                if i != len(head_layer_list) - 1:
                    layers.append(nn.GELU())
            self.head = nn.Sequential(*layers)
            if not model_path:  # synthetic run: portable head weights too (with a checkpoint: torch's default init, as :915-926)
                hw = synth.make_head_weights(self.feature_dim, head_layer_list, mode=init, seed=seed)
                self.head.load_state_dict({k: torch.from_numpy(v) for k, v in hw.items()}, strict=True)
            self.feature_dim = head_layer_list[-1]

    def freeze_backbone(self, unfreeze_layers=()):
        """models.py:929-935: a backbone parameter trains iff one of ``unfreeze_layers`` is a SUBSTRING of its name - whole blocks
        (``"blocks.11"``), parts of one (``"blocks.11.attn"``, ``"blocks.10.mlp.fc2.weight"``) or the final ``"norm"``.
        The backward of this build runs from the loss down to the FIRST block that holds a trainable tensor and stops there, so
        the tensors below every block - ``patch_embed.*``, ``pos_embed``, ``cls_token`` - cannot be trained: asking for them raises
        here, at construction, instead of silently leaving them without gradients."""
        for name, param in self.backbone.named_parameters():
            param.requires_grad = any(u in name for u in unfreeze_layers)
            if param.requires_grad and not (name.startswith("blocks.") or name.startswith("norm.")):
                raise NotImplementedError(f"unfreeze_layers={list(unfreeze_layers)} selects '{name}': gradients of the patch embedding, the "
                                          "position embedding and the cls token are not built (the backward stops at the first trainable block)")
            # frozen tensors never change (no optimizer step, no EMA - they are the EMA's source): derived copies of them, such as the
            # bf16 planes of the plane-GEMM modes, may be cached (engine.weight_planes re-checks requires_grad, storage and version)
            param._tt_static = not param.requires_grad

    def trainable_block_ids(self) -> List[int]:
        """Blocks that hold at least one trainable tensor.  A partly unfrozen block (``"blocks.11.attn"``, as the reference's
        substring match allows) is on the backward path as a whole: all of its gradients are computed, those of its frozen tensors
        are dropped before they reach autograd / the optimizer / the gradient exchange."""
        return [i for i, blk in enumerate(self.backbone.blocks) if any(p.requires_grad for p in blk.parameters())]

    # -- forward surface ---------------------------------------------------------------------------
    @torch.no_grad()
    def get_features(self, input):
        """models.py:965-969: final-norm'd patch tokens (cls dropped) and the last block's attention (always detached, :969)."""
        x = self.backbone._check(input)
        tok, probs = engine.vit_tokens(self.backbone, x, last_block_probs=self.return_attention)
        Fr, N, D = tok.shape
        feats = ops.layernorm_fwd(tok, self.backbone.norm.weight, self.backbone.norm.bias, drop_first_token=True)
        return feats.view(Fr, N - 1, D), probs

    def forward(self, x, use_head=True):
        """models.py:1070-1078.  As in the reference the returned features carry grad whenever autograd is enabled and a
        parameter on their path requires it (the head, the unfrozen blocks): the forward then keeps the activations of the
        trainable part and ``features.backward(...)`` runs the HIP backward kernels (``_ExtractorFunction``).  Under
        ``torch.no_grad()`` - every call on the training path of ``TimeT`` - nothing is kept.  ``attentions`` never carry
        grad (:969)."""
        params = [p for p in (self.parameters() if use_head else self.backbone.parameters()) if p.requires_grad]
        if torch.is_grad_enabled() and params:
            if x.requires_grad:
                raise NotImplementedError("gradients with respect to the input frames are not built (the reference's callers never ask for them)")
            feats, attn = _ExtractorFunction.apply(self, x, bool(use_head), *params)
            return feats, (attn if attn.numel() else None)
        with torch.no_grad():
            feats, attn = self.get_features(x)
            if self.head is not None and use_head:
                Fr, n, D = feats.shape
                feats = engine.head_forward(feats.view(Fr * n, D), self.head).view(Fr, n, -1)
            return feats, attn

    def _forward_saving(self, x, use_head: bool):
        """Forward that keeps what ``_backward_saved`` needs: activations of the trainable blocks (all frames), the final
        norm's statistics and the head's activations."""
        vit = self.backbone
        xi = vit._check(x)
        ids = self.trainable_block_ids()
        first = min(ids) if ids else None
        save = {i: {} for i in range(first, len(vit.blocks))} if first is not None else None
        tok, probs = engine.vit_tokens(vit, xi, None, save, last_block_probs=self.return_attention)
        Fr, N, D = tok.shape
        feats, mean_f, rstd_f = ops.layernorm_fwd(tok, vit.norm.weight, vit.norm.bias, save_stats=True, drop_first_token=True)
        sv_head = None
        out = feats
        if self.head is not None and use_head:
            sv_head = {}
            out = engine.head_forward(feats, self.head, sv_head)
        saved = dict(first=first, blocks=save, tok=tok, mean=mean_f, rstd=rstd_f, head=sv_head, shape=(Fr, N, D))
        return out.view(Fr, N - 1, -1), probs, saved

    def _backward_saved(self, saved: dict, dfeats: torch.Tensor) -> Dict[nn.Parameter, torch.Tensor]:
        vit = self.backbone
        Fr, N, D = saved["shape"]
        grads: Dict[nn.Parameter, torch.Tensor] = {}
        d = dfeats.reshape(Fr * (N - 1), -1).contiguous()   # read-only below: autograd's buffer is never written
        if saved["head"] is not None:
            d = engine.head_backward(d, self.head, saved["head"], grads)
        first = saved["first"]
        wg = vit.norm.weight.requires_grad
        if first is not None or wg:
            dx, dg, db = ops.layernorm_bwd(d, saved["tok"], vit.norm.weight, saved["mean"], saved["rstd"], need_wgrad=wg, drop_first_token=True)
            if wg:
                grads[vit.norm.weight], grads[vit.norm.bias] = dg, db
            if first is not None:
                dx = dx.view(Fr * N, D)
                for i in range(len(vit.blocks) - 1, first - 1, -1):
                    dx = engine.block_backward(dx, vit.blocks[i], vit.num_heads, saved["blocks"][i], 0, Fr, grads, need_dx=i > first)
        return {p: g for p, g in grads.items() if p.requires_grad}


class _ExtractorFunction(torch.autograd.Function):
    """``FeatureExtractor.forward`` with grad: HIP forward that keeps the trainable part's activations, HIP backward
    (engine.head_backward / block_backward) when the features' gradient arrives."""

    @staticmethod
    def forward(ctx, fe, x, use_head, *params):
        feats, attn, saved = fe._forward_saving(x, use_head)
        ctx.fe, ctx.saved, ctx.params = fe, saved, params
        if attn is None:
            attn = x.new_empty(0)   # autograd.Function outputs must be tensors; turned back into None by the caller-facing wrapper
        ctx.mark_non_differentiable(attn)
        return feats, attn

    @staticmethod
    def backward(ctx, dfeats, _dattn):
        if ctx.saved is None:
            raise RuntimeError("FeatureExtractor: backward through the same forward twice (activations are freed after the first)")
        grads = ctx.fe._backward_saved(ctx.saved, dfeats)
        ctx.saved = None
        return (None, None, None, *[grads.get(p) for p in ctx.params])


class _DDPStandIn(nn.Module):
    """Stands where ``torch.nn.parallel.DistributedDataParallel`` stands in the reference: one attribute, ``module``, so
    that the wrapper's state_dict keys read ``model.module.<...>`` exactly as the reference's multi-GPU checkpoints do."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *input):
        return self.module(*input)


class DistributedDataParallelModel(nn.Module):
    """``models.py:1292-1306`` surface (``forward``, ``get_non_ddp_model``, attribute fall-through, state_dict keys
    ``model.module.*``) without ``torch.nn.parallel.DistributedDataParallel``: the step's backward already produces every
    gradient in one shot, so the exchange is the bucketed RCCL all-reduce issued by ``TimeT`` itself (engine.GradExchange), and
    the Sinkhorn solve all-gathers the score rows (engine.global_sinkhorn).  Parameters and buffers are broadcast from rank 0
    at construction, as DDP does."""

    def __init__(self, model, gpu):
        super().__init__()
        self.model = _DDPStandIn(model)
        self.gpu = gpu
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0)
            model.data_parallel = True

    def forward(self, *input):
        return self.model(*input)

    def get_non_ddp_model(self):
        return self.model.module

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("model").module, name)
