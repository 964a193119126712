"""Checkpoint handling: tensor inventory, safetensors reader, seeded random initialisation.

Tensor names follow the HF-hub ``rednote-hilab/dots.ocr`` state dict that the reference
loads at dots_ocr/parser.py:68-74 (``vision_tower.*`` for the NaViT encoder + patch merger,
``model.*`` / ``lm_head.*`` for the Qwen2 LM).  Norm type and bias presence are decided by
the state dict itself (a norm with a ``.bias`` is a LayerNorm), never hard-coded.
"""
from __future__ import annotations

import json
import struct
from pathlib import Path
from typing import Dict, Iterator, Tuple

import numpy as np
import torch

from .config import DotsConfig


def expected_tensors(cfg: DotsConfig) -> Dict[str, Tuple[int, ...]]:
    v = cfg.vision
    t: Dict[str, Tuple[int, ...]] = {}
    E, I = v.embed_dim, v.intermediate_size
    t["vision_tower.patch_embed.patchifier.proj.weight"] = (E, v.num_channels, v.patch_size, v.patch_size)
    t["vision_tower.patch_embed.patchifier.proj.bias"] = (E,)
    t["vision_tower.patch_embed.patchifier.norm.weight"] = (E,)
    for i in range(v.num_hidden_layers):
        p = f"vision_tower.blocks.{i}."
        t[p + "norm1.weight"] = (E,)
        t[p + "attn.qkv.weight"] = (3 * E, E)
        t[p + "attn.proj.weight"] = (E, E)
        t[p + "norm2.weight"] = (E,)
        t[p + "mlp.fc1.weight"] = (I, E)
        t[p + "mlp.fc2.weight"] = (E, I)
        t[p + "mlp.fc3.weight"] = (I, E)
        if v.use_bias:
            t[p + "attn.qkv.bias"] = (3 * E,)
            t[p + "attn.proj.bias"] = (E,)
            t[p + "mlp.fc1.bias"] = (I,)
            t[p + "mlp.fc2.bias"] = (E,)
            t[p + "mlp.fc3.bias"] = (I,)
    if v.post_norm:
        t["vision_tower.post_trunk_norm.weight"] = (E,)
    M = E * v.spatial_merge_size ** 2
    t["vision_tower.merger.ln_q.weight"] = (E,)
    t["vision_tower.merger.ln_q.bias"] = (E,)
    t["vision_tower.merger.mlp.0.weight"] = (M, M)
    t["vision_tower.merger.mlp.0.bias"] = (M,)
    t["vision_tower.merger.mlp.2.weight"] = (v.hidden_size, M)
    t["vision_tower.merger.mlp.2.bias"] = (v.hidden_size,)

    H, Iq = cfg.hidden_size, cfg.intermediate_size
    t["model.embed_tokens.weight"] = (cfg.vocab_size, H)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        t[p + "input_layernorm.weight"] = (H,)
        t[p + "self_attn.q_proj.weight"] = (cfg.q_size, H)
        t[p + "self_attn.k_proj.weight"] = (cfg.kv_size, H)
        t[p + "self_attn.v_proj.weight"] = (cfg.kv_size, H)
        if cfg.attention_bias:
            t[p + "self_attn.q_proj.bias"] = (cfg.q_size,)
            t[p + "self_attn.k_proj.bias"] = (cfg.kv_size,)
            t[p + "self_attn.v_proj.bias"] = (cfg.kv_size,)
        t[p + "self_attn.o_proj.weight"] = (H, cfg.q_size)
        t[p + "post_attention_layernorm.weight"] = (H,)
        t[p + "mlp.gate_proj.weight"] = (Iq, H)
        t[p + "mlp.up_proj.weight"] = (Iq, H)
        t[p + "mlp.down_proj.weight"] = (H, Iq)
    t["model.norm.weight"] = (H,)
    if not cfg.tie_word_embeddings:
        t["lm_head.weight"] = (cfg.vocab_size, H)
    return t


def _is_norm_scale(name: str) -> bool:
    return name.endswith(("norm.weight", "norm1.weight", "norm2.weight", "layernorm.weight", "ln_q.weight"))


def random_state_dict(cfg: DotsConfig, seed: int = 0, std: float = 0.02,
                      dtype: torch.dtype = torch.bfloat16, threads: int = 1) -> Dict[str, torch.Tensor]:
    """Seeded N(0, std) weights at the checkpoint's shapes; norm scales ~1, biases small.
    Each tensor gets its own generator keyed by (seed, index) so the result does not depend on
    thread scheduling and a single tensor can be regenerated in isolation."""
    items = list(expected_tensors(cfg).items())

    def make(idx_name_shape):
        idx, (name, shape) = idx_name_shape
        g = torch.Generator().manual_seed((seed * 1000003 + idx * 7919 + 17) % (2 ** 31))
        if _is_norm_scale(name):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            w = std * torch.randn(shape, generator=g)
        return name, w.to(dtype)

    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as ex:
            return dict(ex.map(make, enumerate(items)))
    return dict(map(make, enumerate(items)))


# ---------------------------------------------------------------------- safetensors reader
_ST_DTYPES = {"BF16": (torch.bfloat16, 2), "F16": (torch.float16, 2), "F32": (torch.float32, 4)}


def iter_safetensors(path: Path) -> Iterator[Tuple[str, torch.Tensor]]:
    """Minimal zero-dependency safetensors reader (mmap + header JSON)."""
    with open(path, "rb") as f:
        (hlen,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hlen))
    base = 8 + hlen
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        dt, _ = _ST_DTYPES[meta["dtype"]]
        s, e = meta["data_offsets"]
        buf = torch.from_numpy(np.asarray(mm[base + s: base + e]))
        yield name, buf.view(dt).reshape(meta["shape"])


def load_state_dict(model_dir: str | Path) -> Dict[str, torch.Tensor]:
    model_dir = Path(model_dir)
    files = sorted(model_dir.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {model_dir}")
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        for name, t in iter_safetensors(f):
            sd[name] = t
    return sd


def save_safetensors(sd: Dict[str, torch.Tensor], path: Path) -> None:
    """Writer used by tests to round-trip the reader (bf16/f16/f32 only)."""
    inv = {torch.bfloat16: "BF16", torch.float16: "F16", torch.float32: "F32"}
    header, off, blobs = {}, 0, []
    for name, t in sd.items():
        t = t.contiguous()
        raw = t.view(torch.uint8).numpy().tobytes() if t.dtype != torch.bfloat16 \
            else t.view(torch.int16).numpy().tobytes()
        header[name] = {"dtype": inv[t.dtype], "shape": list(t.shape), "data_offsets": [off, off + len(raw)]}
        off += len(raw)
        blobs.append(raw)
    hj = json.dumps(header).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)
