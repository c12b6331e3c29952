"""DINO ViT whose arithmetic runs in libtimetuning_hip.so.

The module tree only *holds parameters* under the names the reference's ViT uses
(``dino_vision_transformer.py:174-199``: ``cls_token``, ``pos_embed``, ``patch_embed.proj``,
``blocks.{i}.norm1|attn.qkv|attn.proj|norm2|mlp.fc1|mlp.fc2``, ``norm``), so reference
checkpoints load unchanged.  The torch sub-modules are never called: every public method
launches HIP kernels through ``engine`` / ``hip_ops`` and raises if the library is missing.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import engine
from . import hip_ops as ops


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the HIP path does not call torch sub-modules")


class Mlp(_NoForward):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Attention(_NoForward):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(_NoForward):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed(_NoForward):
    def __init__(self, img_size: int = 224, patch_size: int = 16, in_chans: int = 3, embed_dim: int = 768):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    """Same constructor surface as the reference class for the arguments its factories use
    (``dino_vision_transformer.py:276-294``); drop rates are fixed at 0 as there."""

    def __init__(self, img_size=(224,), patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, **kwargs):
        super().__init__()
        if embed_dim // num_heads != 64:
            raise ValueError("the fused attention kernel is built for head_dim 64 (ViT-S/B); got "
                             f"{embed_dim // num_heads}")
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.patch_embed = PatchEmbed(img_size[0], patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Identity()
        # init mirrors dino_vision_transformer.py:201-212 (trunc_normal 0.02, LN 1/0, zero biases)
        with torch.no_grad():
            nn.init.trunc_normal_(self.pos_embed, std=0.02)
            nn.init.trunc_normal_(self.cls_token, std=0.02)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=0.02)
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)

    # -- the reference's public methods ---------------------------------------------------------
    def _check(self, x: torch.Tensor):
        P = self.patch_embed.patch_size
        if x.shape[-1] % P or x.shape[-2] % P or x.shape[-1] < P or x.shape[-2] < P:
            raise ValueError(f"input height and width must be positive multiples of the patch size {P} (got {tuple(x.shape[-2:])})")
        return x.contiguous().float()

    def pos_table(self, H: int, W: int) -> torch.Tensor:
        """``interpolate_pos_encoding`` (dino_vision_transformer.py:214-234) as a [1 + rows*cols, D] table: the stored
        ``pos_embed`` when the token grid is the stored square one, else its bicubic resampling (computed once per input size
        and parameter version)."""
        P, D = self.patch_embed.patch_size, self.embed_dim
        rows, cols = H // P, W // P
        if rows * cols == self.pos_embed.shape[1] - 1 and rows == cols:
            return self.pos_embed.view(-1, D)
        key = (rows, cols, self.pos_embed.data_ptr(), self.pos_embed._version)
        cache = self.__dict__.setdefault("_pos_cache", {})
        if key not in cache:
            cache.clear()
            with torch.no_grad():
                cache[key] = ops.pos_embed_interpolate(self.pos_embed.detach().view(-1, D).contiguous(), rows, cols)
        return cache[key]

    def prepare_tokens(self, x: torch.Tensor) -> torch.Tensor:
        pe = self.patch_embed.proj
        D = self.embed_dim
        x = self._check(x)
        return ops.patch_embed_fwd(x, pe.weight.view(D, -1), pe.bias, self.cls_token.view(D), self.pos_table(x.shape[-2], x.shape[-1]),
                                   self.patch_embed.patch_size)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.get_intermediate_layers(x, 1)[0][:, 0]

    @torch.no_grad()
    def get_last_selfattention(self, x: torch.Tensor) -> torch.Tensor:
        _, probs = engine.vit_tokens(self, self._check(x), last_block_probs=True)
        return probs

    @torch.no_grad()
    def get_intermediate_layers(self, x: torch.Tensor, n: int = 1) -> List[torch.Tensor]:
        if n != 1:
            raise NotImplementedError("the training path only uses n=1 (models.py:966)")
        tok, _ = engine.vit_tokens(self, self._check(x))
        return [ops.layernorm_fwd(tok, self.norm.weight, self.norm.bias)]


def vit_small(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True, **kw)


def vit_base(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, **kw)
