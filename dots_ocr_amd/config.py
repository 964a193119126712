"""Model configuration for the dots.ocr engine.

Every dimension is read from the checkpoint's ``config.json`` / ``preprocessor_config.json``
at load time (reference: dots_ocr/parser.py:67-75 loads them through
``AutoModelForCausalLM.from_pretrained(..., trust_remote_code=True)``).  The defaults below
are the values SURVEY.md §8(a) recalls for rednote-hilab/dots.ocr and are only used to build
random-weight models for tests and benchmarks when no checkpoint is present.
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field
from pathlib import Path


@dataclass
class VisionConfig:
    embed_dim: int = 1536
    num_hidden_layers: int = 42
    num_attention_heads: int = 12
    intermediate_size: int = 4224
    patch_size: int = 14
    spatial_merge_size: int = 2
    temporal_patch_size: int = 1
    num_channels: int = 3
    rms_norm_eps: float = 1e-5
    merger_ln_eps: float = 1e-6
    use_bias: bool = False
    post_norm: bool = True
    hidden_size: int = 1536          # merger output dim == LM hidden size

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_attention_heads

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.temporal_patch_size * self.patch_size * self.patch_size


@dataclass
class DotsConfig:
    # language model (Qwen2 architecture)
    hidden_size: int = 1536
    num_hidden_layers: int = 28
    num_attention_heads: int = 12
    num_key_value_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 8960
    vocab_size: int = 151936
    rope_theta: float = 1000000.0
    rms_norm_eps: float = 1e-6
    attention_bias: bool = True
    tie_word_embeddings: bool = False
    max_position_embeddings: int = 131072
    image_token_id: int = 151665
    eos_token_ids: tuple = (151643, 151673)
    pad_token_id: int = 151643
    vision: VisionConfig = field(default_factory=VisionConfig)
    # preprocessor
    min_pixels: int = 3136
    max_pixels: int = 11289600
    image_mean: tuple = (0.48145466, 0.4578275, 0.40821073)
    image_std: tuple = (0.26862954, 0.26130258, 0.27577711)

    # ------------------------------------------------------------------ constructors
    @staticmethod
    def tiny(layers: int = 2, v_layers: int = 2, vocab: int = 1024) -> "DotsConfig":
        """Small-dims variant for fast tests.  Keeps what the kernels specialise on:
        head_dim 128, GQA group 6 (q heads 6 : kv heads 1), qkv bias, untied lm_head,
        patch 14 / merge 2, vision heads of 128."""
        return DotsConfig(
            hidden_size=768, num_hidden_layers=layers, num_attention_heads=6, num_key_value_heads=1,
            head_dim=128, intermediate_size=1536, vocab_size=vocab, image_token_id=vocab - 3,
            eos_token_ids=(vocab - 1, vocab - 2), pad_token_id=vocab - 1,
            vision=VisionConfig(embed_dim=256, num_hidden_layers=v_layers, num_attention_heads=2,
                                intermediate_size=512, hidden_size=768),
        )

    @staticmethod
    def from_pretrained(path: str | Path) -> "DotsConfig":
        path = Path(path)
        cfg = json.loads((path / "config.json").read_text())
        v = cfg.get("vision_config", {})
        vision = VisionConfig(
            embed_dim=v.get("embed_dim", 1536),
            num_hidden_layers=v.get("num_hidden_layers", 42),
            num_attention_heads=v.get("num_attention_heads", 12),
            intermediate_size=v.get("intermediate_size", 4224),
            patch_size=v.get("patch_size", 14),
            spatial_merge_size=v.get("spatial_merge_size", 2),
            temporal_patch_size=v.get("temporal_patch_size", 1),
            num_channels=v.get("num_channels", 3),
            rms_norm_eps=v.get("rms_norm_eps", 1e-5),
            use_bias=v.get("use_bias", False),
            post_norm=v.get("post_norm", True),
            hidden_size=v.get("hidden_size", cfg["hidden_size"]),
        )
        heads = cfg["num_attention_heads"]
        eos = cfg.get("eos_token_id", 151643)
        gen = path / "generation_config.json"
        if gen.exists():
            eos = json.loads(gen.read_text()).get("eos_token_id", eos)
        if isinstance(eos, int):
            eos = [eos]
        out = DotsConfig(
            hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
            num_attention_heads=heads, num_key_value_heads=cfg.get("num_key_value_heads", heads),
            head_dim=cfg.get("head_dim", cfg["hidden_size"] // heads),
            intermediate_size=cfg["intermediate_size"], vocab_size=cfg["vocab_size"],
            rope_theta=float(cfg.get("rope_theta", 1e6)), rms_norm_eps=float(cfg.get("rms_norm_eps", 1e-6)),
            attention_bias=cfg.get("attention_bias", True),
            tie_word_embeddings=cfg.get("tie_word_embeddings", False),
            max_position_embeddings=cfg.get("max_position_embeddings", 131072),
            image_token_id=cfg.get("image_token_id", 151665),
            eos_token_ids=tuple(eos), pad_token_id=cfg.get("pad_token_id", eos[0]) or eos[0],
            vision=vision,
        )
        pp = path / "preprocessor_config.json"
        if pp.exists():
            p = json.loads(pp.read_text())
            out.min_pixels = p.get("min_pixels", out.min_pixels)
            out.max_pixels = p.get("max_pixels", out.max_pixels)
            out.image_mean = tuple(p.get("image_mean", out.image_mean))
            out.image_std = tuple(p.get("image_std", out.image_std))
        return out

    def to_dict(self) -> dict:
        return asdict(self)

    # ------------------------------------------------------------------ derived sizes
    @property
    def q_size(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_key_value_heads * self.head_dim

    def lm_param_count(self) -> int:
        h, i = self.hidden_size, self.intermediate_size
        per = h * (self.q_size + 2 * self.kv_size) + (self.q_size + 2 * self.kv_size if self.attention_bias else 0) \
            + self.q_size * h + 3 * h * i + 2 * h
        n = self.num_hidden_layers * per + h + self.vocab_size * h
        if not self.tie_word_embeddings:
            n += self.vocab_size * h
        return n
