"""Portable synthetic weights and clips.

There is no network on the build or GPU boxes, so the pretrained DINO checkpoint the
reference pulls through ``torch.hub`` (reference ``models.py:780-785``) cannot be used.
Every tensor is instead drawn from a counter-based generator (numpy Philox) keyed by the
tensor's state_dict name, so that the golden-vector generator (which imports the
reference), the CPU oracle, the tests and ``bench.py`` all regenerate bit-identical
weights without shipping 87 MB of parameters.

Two initialisation modes:

``dino``    mirrors ``dino_vision_transformer.py:201-212``: ``trunc_normal(std=0.02)`` for
            Linear / conv / pos_embed / cls_token, LayerNorm gamma=1 beta=0, biases 0.
``stress``  same shapes, but non-zero biases, non-unit LayerNorm gains and a larger
            weight std, so that a kernel which drops a bias, a gain or mis-scales the
            attention logits cannot pass a parity test by accident.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np

# (embed_dim, depth, heads, patch) per architecture name used by the reference
# (``models.py:780-785`` hub entries; ``dino_vision_transformer.py:276-294`` factories).
ARCHS = {
    "dino-s16": dict(embed_dim=384, depth=12, num_heads=6, patch_size=16),
    "dino-s8": dict(embed_dim=384, depth=12, num_heads=6, patch_size=8),
    "dino-b16": dict(embed_dim=768, depth=12, num_heads=12, patch_size=16),
    # not a reference architecture: a cheap ViT for CPU-sized parity tests.  It is only
    # reachable through explicit ``vit_cfg=`` arguments, never through the CLI.
    "tiny-s16": dict(embed_dim=128, depth=12, num_heads=2, patch_size=16),
}


def _philox(name: str, seed: int) -> np.random.Generator:
    digest = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = np.frombuffer(digest[:16], dtype=np.uint64)
    return np.random.Generator(np.random.Philox(key=key))


def normal(name: str, shape, std: float = 1.0, mean: float = 0.0, seed: int = 1) -> np.ndarray:
    """fp32 normal tensor that depends only on (name, seed, shape)."""
    g = _philox(name, seed)
    a = g.standard_normal(size=tuple(shape), dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return a


def make_sinkhorn_w8_scores() -> np.ndarray:
    """Scores of the ``sinkhorn_w8.npz`` fixture (tests/golden; REGENERATED, 53 MB, not stored): [8 * 8320, 200] fp32 - SURVEY 8(c)'s
    third Sinkhorn case (K, B_loc, iters, W) = (200, 8320, 10, 8), BASELINE config C3's global problem (6272 patches + 2048 queue
    rows per rank).  Integer factors with |entry| <= 8 over 32 terms: every dot product is an integer below 2^12, so the fp32
    matmul and the division by 1024 are exact on any BLAS and a test's scores equal the generator's bit for bit."""
    xi = np.clip(np.rint(4.0 * normal("skw8.x", (8 * 8320, 32))), -8, 8).astype(np.float32)
    pi = np.clip(np.rint(4.0 * normal("skw8.p", (200, 32))), -8, 8).astype(np.float32)
    return (xi @ pi.T) / np.float32(1024.0)


def make_scaler_features() -> np.ndarray:
    """Input of the ``scaler.npz`` fixture (tests/golden; regenerated, not stored): 230 000 x 6 fp32 - three batches of the
    reference's 100 000-row ``partial_fit`` loop (my_utils.py:23-30) - with unequal column scales / offsets and one constant
    column (zero variance: scikit-learn's scale 1)."""
    x = normal("scaler.x", (230000, 6))
    x *= np.array([1.0, 0.02, 35.0, 1.0, 0.0, 3.0], np.float32)
    x += np.array([0.0, 5.0, -120.0, 1e3, 2.5, 0.0], np.float32)
    return x


def vit_param_shapes(embed_dim: int, depth: int, num_heads: int, patch_size: int,
                     img_size: int = 224, in_chans: int = 3, mlp_ratio: int = 4) -> "OrderedDict[str, tuple]":
    """state_dict layout of the DINO ViT (``dino_vision_transformer.py:174-199``)."""
    n = (img_size // patch_size) ** 2
    D = embed_dim
    H = D * mlp_ratio
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["cls_token"] = (1, 1, D)
    s["pos_embed"] = (1, n + 1, D)
    s["patch_embed.proj.weight"] = (D, in_chans, patch_size, patch_size)
    s["patch_embed.proj.bias"] = (D,)
    for i in range(depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (H, D)
        s[p + "mlp.fc1.bias"] = (H,)
        s[p + "mlp.fc2.weight"] = (D, H)
        s[p + "mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,)
    s["norm.bias"] = (D,)
    return s


def head_param_shapes(in_dim: int, head_layer_list) -> "OrderedDict[str, tuple]":
    """``nn.Sequential`` indices 0,2,4,.. are the Linears (``models.py:915-926``)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    prev = in_dim
    for i, width in enumerate(head_layer_list):
        s[f"{2 * i}.weight"] = (width, prev)
        s[f"{2 * i}.bias"] = (width,)
        prev = width
    return s


def _fill(name: str, shape, mode: str, seed: int) -> np.ndarray:
    is_norm = ".norm" in name or name.startswith("norm")
    if is_norm and name.endswith("weight"):
        if mode == "dino":
            return np.ones(shape, np.float32)
        return normal(name, shape, 0.1, 1.0, seed)
    if name.endswith("bias"):
        if mode == "dino":
            return np.zeros(shape, np.float32)
        return normal(name, shape, 0.05, 0.0, seed)
    std = 0.02 if mode == "dino" else 0.06
    return np.clip(normal(name, shape, std, 0.0, seed), -2.0, 2.0)


def make_vit_weights(prefix: str = "", mode: str = "dino", seed: int = 1, **cfg) -> "OrderedDict[str, np.ndarray]":
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shp in vit_param_shapes(**cfg).items():
        out[prefix + k] = _fill("backbone." + k, shp, mode, seed)
    return out


def make_head_weights(in_dim: int, head_layer_list, prefix: str = "", mode: str = "dino",
                      seed: int = 1) -> "OrderedDict[str, np.ndarray]":
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shp in head_param_shapes(in_dim, head_layer_list).items():
        if k.endswith("weight"):
            fan_in = shp[1]
            # nn.Linear's default scale (uniform(+-1/sqrt(fan_in))) has std 1/sqrt(3 fan_in)
            out[prefix + k] = normal("head." + k, shp, (1.0 / (3.0 * fan_in)) ** 0.5, 0.0, seed)
        else:
            out[prefix + k] = normal("head." + k, shp, 0.02 if mode == "dino" else 0.05, 0.0, seed)
    return out


def make_prototypes(num_prototypes: int, dim: int, seed: int = 1) -> np.ndarray:
    """``normalize(randn(K, dim))`` (``time_tuning.py:90-93``) from the portable stream."""
    p = normal("prototypes", (num_prototypes, dim), 1.0, 0.0, seed)
    p /= np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-12).astype(np.float32)
    return p.astype(np.float32)


def make_clips(bs: int, fs: int, img: int = 224, seed: int = 1, coherent: bool = True,
               name: str = "clips") -> np.ndarray:
    """Synthetic normalised clips ``[bs, fs, 3, img, img]`` fp32.

    The real loader yields ImageNet-normalised pixels (``time_tuning.py:592``), i.e. roughly
    unit normal.  ``coherent=True`` makes frame t a (2t, 2t)-pixel roll of frame 0 plus a
    little noise so that label propagation has real temporal structure to follow.
    """
    if not coherent:
        return normal(name, (bs, fs, 3, img, img), 1.0, 0.0, seed)
    base = normal(name + ".base", (bs, 1, 3, img, img), 1.0, 0.0, seed)
    noise = normal(name + ".noise", (bs, fs, 3, img, img), 0.05, 0.0, seed)
    frames = [np.roll(base[:, 0], shift=(2 * t, 2 * t), axis=(-2, -1)) for t in range(fs)]
    return (np.stack(frames, axis=1) + noise).astype(np.float32)


def make_smooth_clips(bs: int, fs: int, img: int = 224, seed: int = 1, grid: int = 5, amplitude: float = 1.5,
                      noise_std: float = 0.02, name: str = "smooth_clips") -> np.ndarray:
    """Clips whose frames are LOW-FREQUENCY fields (a ``grid x grid`` normal lattice per channel, bilinearly
    interpolated to ``img x img``) plus a little noise; frame t is a (4t, 4t)-pixel roll of frame 0.

    Natural frames give the ViT spatially smooth cls-attention; white-noise frames (``make_clips``) give speckled
    attention.  The ``--use_mask`` fixtures need the former: the reference's ``process_attentions`` raises on any
    frame whose thresholded attention has a component of <= 2 pixels (see oracle/gen_golden.py)."""
    lattice = normal(name + ".lattice", (bs, 3, grid, grid), amplitude, 0.0, seed).astype(np.float64)
    pos = np.linspace(0.0, grid - 1.0, img)
    i0 = np.minimum(np.floor(pos).astype(np.int64), grid - 2)
    w = pos - i0
    rows = lattice[:, :, i0, :] * (1.0 - w)[None, None, :, None] + lattice[:, :, i0 + 1, :] * w[None, None, :, None]
    base = rows[:, :, :, i0] * (1.0 - w) + rows[:, :, :, i0 + 1] * w                       # [bs,3,img,img]
    noise = normal(name + ".noise", (bs, fs, 3, img, img), noise_std, 0.0, seed)
    frames = [np.roll(base, shift=(4 * t, 4 * t), axis=(-2, -1)) for t in range(fs)]
    return (np.stack(frames, axis=1) + noise).astype(np.float32)
