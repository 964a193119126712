"""Synthetic document pages and prompts for benchmarks and smoke tests:
white RGB canvases with black text lines drawn with PIL's default font, seeded per page."""
from __future__ import annotations

import random
import string
from typing import List, Tuple

import numpy as np
from PIL import Image, ImageDraw

from .config import DotsConfig

A4_200DPI = (1654, 2339)        # width, height -> smart_resize 1652 x 2352 -> 118 x 168 patches -> 4956 vision tokens
HIGH_RES = (1344, 1344)         # BASELINE config 3 ("3x3 tiles" restated): 96 x 96 patches -> 2304 vision tokens


def synth_page(index: int, size: Tuple[int, int] = A4_200DPI, seed: int = 1234) -> Image.Image:
    rng = random.Random(seed + index)
    w, h = size
    img = Image.new("RGB", (w, h), (255, 255, 255))
    d = ImageDraw.Draw(img)
    n_lines = rng.randint(40, 60)
    alphabet = string.ascii_letters + string.digits + "     .,;:-"
    for i in range(n_lines):
        y = int((i + 1) * h / (n_lines + 2))
        x = rng.randint(w // 20, w // 8)
        text = "".join(rng.choice(alphabet) for _ in range(rng.randint(30, max(31, w // 12))))
        d.text((x, y), text, fill=(0, 0, 0))
    return img


def synth_prompt_ids(cfg: DotsConfig, n_vision_tokens: int, n_text_tokens: int = 241, seed: int = 0) -> np.ndarray:
    """<3 chat tokens> <n image pads> <n_text prompt tokens>: the shape of the reference's chat template
    around `prompt_layout_all_en` (about 240 BPE tokens) when no checkpoint tokenizer is available."""
    rng = np.random.default_rng(seed)
    hi = min(cfg.vocab_size, cfg.image_token_id) - 1
    head = rng.integers(0, hi, 3)
    tail = rng.integers(0, hi, n_text_tokens)
    return np.concatenate([head, np.full(n_vision_tokens, cfg.image_token_id), tail]).astype(np.int32)
