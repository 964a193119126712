"""Processor object honouring the three HF calls the reference makes (dots_ocr/parser.py:93-116):
``apply_chat_template``, ``__call__(text=, images=, padding=, return_tensors=)`` and ``batch_decode``,
plus ``process_vision_info`` (qwen_vl_utils, parser.py:98).

Tokenizer: the checkpoint's ``tokenizer.json`` through the `tokenizers` library when the weights
directory is present.  Without a checkpoint (this repo ships none and has no network) a byte-level
stand-in with the same special-token roles keeps the whole pipeline runnable on random weights —
it is only ever used together with random weights and says so in its name.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from .config import DotsConfig
from .image_utils import fetch_image, preprocess_image

IMG_START, IMG_PAD, IMG_END = "<|img|>", "<|imgpad|>", "<|endofimg|>"      # spellings: reference model/inference.py:33
USER, END_USER, ASSISTANT, END_ASSISTANT = "<|user|>", "<|endofuser|>", "<|assistant|>", "<|endofassistant|>"


class BatchFeature(dict):
    """dict with attribute access and .to(device), like transformers.BatchFeature."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        import torch
        return BatchFeature({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})


class SyntheticByteTokenizer:
    """Stand-in used only when no checkpoint tokenizer exists: UTF-8 bytes are ids 0..255, the chat /
    image special tokens sit just below the vocabulary end (image pad == config.image_token_id)."""

    def __init__(self, cfg: DotsConfig):
        self.cfg = cfg
        top = cfg.vocab_size - 1
        reserved = {cfg.image_token_id, *cfg.eos_token_ids}
        self.special = {IMG_PAD: cfg.image_token_id, "<|endoftext|>": cfg.eos_token_ids[0]}
        if len(cfg.eos_token_ids) > 1:
            self.special[END_ASSISTANT] = cfg.eos_token_ids[1]
        for tok in (USER, END_USER, ASSISTANT, END_ASSISTANT, IMG_START, IMG_END):
            if tok in self.special:
                continue
            while top in reserved or top < 256:
                top -= 1
            self.special[tok] = top
            reserved.add(top)
        self.inv_special = {v: k for k, v in self.special.items()}
        self.pad_token_id = cfg.pad_token_id

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        i = 0
        while i < len(text):
            if text[i] == "<":
                hit = next((t for t in self.special if text.startswith(t, i)), None)
                if hit:
                    out.append(self.special[hit])
                    i += len(hit)
                    continue
            out.extend(text[i].encode("utf-8"))
            i += 1
        return out

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        parts, buf = [], bytearray()
        for t in ids:
            t = int(t)
            if t < 256:
                buf.append(t)
                continue
            if buf:
                parts.append(buf.decode("utf-8", errors="replace"))
                buf = bytearray()
            if t in self.inv_special:
                if not skip_special_tokens:
                    parts.append(self.inv_special[t])
            else:
                parts.append("" if skip_special_tokens else f"<|{t}|>")
        if buf:
            parts.append(buf.decode("utf-8", errors="replace"))
        return "".join(parts)


class HFJsonTokenizer:
    """The checkpoint's tokenizer.json via the `tokenizers` runtime."""

    def __init__(self, path: Path, cfg: DotsConfig):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(str(path / "tokenizer.json"))
        self.pad_token_id = cfg.pad_token_id
        self.special_tokens_map = {}                 # bos/eos/pad strings a chat template may reference
        for name in ("special_tokens_map.json", "tokenizer_config.json"):
            f = path / name
            if f.exists():
                d = json.loads(f.read_text(encoding="utf-8"))
                for k in ("bos_token", "eos_token", "pad_token"):
                    v = d.get(k)
                    if isinstance(v, dict):
                        v = v.get("content")
                    if isinstance(v, str) and k not in self.special_tokens_map:
                        self.special_tokens_map[k] = v

    def encode(self, text: str) -> List[int]:
        return self.tk.encode(text, add_special_tokens=False).ids

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        return self.tk.decode([int(i) for i in ids], skip_special_tokens=skip_special_tokens)


def process_vision_info(messages):
    """messages -> ([PIL RGB images], None): same contract as qwen_vl_utils.process_vision_info."""
    images = []
    for msg in messages if isinstance(messages[0], dict) else [m for conv in messages for m in conv]:
        content = msg.get("content")
        if not isinstance(content, list):
            continue
        for item in content:
            if item.get("type") == "image" or "image" in item:
                images.append(fetch_image(item["image"], min_pixels=item.get("min_pixels"), max_pixels=item.get("max_pixels"),
                                          resized_height=item.get("resized_height"), resized_width=item.get("resized_width")))
    return (images or None), None


def load_chat_template(path: Path) -> Optional[str]:
    """The places transformers looks for a processor / tokenizer chat template, in its order of precedence."""
    path = Path(path)
    if (path / "chat_template.jinja").exists():
        return (path / "chat_template.jinja").read_text(encoding="utf-8")
    for name in ("chat_template.json", "processor_config.json", "tokenizer_config.json"):
        f = path / name
        if f.exists():
            t = json.loads(f.read_text(encoding="utf-8")).get("chat_template")
            if isinstance(t, str) and t:
                return t
    return None


class DotsOcrProcessor:
    """`engine`: when a dots_ocr_amd.engine.Engine is attached, images are resized / normalised / patchified on its
    GPU (bit-identical to the host Pillow path, ~100x faster for an A4 page) and `pixel_values` comes back as a CUDA
    tensor; without it the host path of image_utils.preprocess_image is used."""

    def __init__(self, cfg: DotsConfig, tokenizer=None, engine=None, chat_template: Optional[str] = None):
        self.cfg = cfg
        self.tokenizer = tokenizer or SyntheticByteTokenizer(cfg)
        self.engine = engine
        self.chat_template = chat_template          # the checkpoint's Jinja template, when it ships one
        self._compiled_template = None

    @classmethod
    def from_pretrained(cls, path, engine=None, **_):
        path = Path(path)
        cfg = DotsConfig.from_pretrained(path)
        tok = HFJsonTokenizer(path, cfg) if (path / "tokenizer.json").exists() else None
        return cls(cfg, tok, engine, load_chat_template(path))

    def _preprocess_on_device(self, images):
        import torch
        from .image_utils import smart_resize, to_rgb
        v = self.cfg.vision
        arrays, grids = [], []
        for im in images:
            a = np.asarray(to_rgb(im), dtype=np.uint8)
            rh, rw = smart_resize(a.shape[0], a.shape[1], v.patch_size * v.spatial_merge_size, self.cfg.min_pixels, self.cfg.max_pixels)
            arrays.append(a)
            grids.append([1, rh // v.patch_size, rw // v.patch_size])
        n = sum(g[1] * g[2] for g in grids)
        pv = torch.empty((n, v.patch_dim), dtype=torch.float32, device=torch.device("cuda", self.engine.device))
        torch.cuda.synchronize(pv.device)
        off = 0
        for a, g in zip(arrays, grids):
            got = self.engine.preprocess_image(a, pv.data_ptr() + off * v.patch_dim * 4)
            assert got == g
            off += g[1] * g[2]
        return pv, grids

    # parser.py:93-97
    def _render_template(self, messages, add_generation_prompt: bool) -> str:
        """Render the checkpoint's Jinja chat template the way transformers does (sandboxed, trim/lstrip blocks,
        `raise_exception`, the special-token variables)."""
        if self._compiled_template is None:
            from jinja2.sandbox import ImmutableSandboxedEnvironment

            def raise_exception(message):
                raise ValueError(message)
            env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
            env.globals["raise_exception"] = raise_exception
            self._compiled_template = env.from_string(self.chat_template)
        special = getattr(self.tokenizer, "special_tokens_map", {}) or {}
        return self._compiled_template.render(messages=messages, add_generation_prompt=add_generation_prompt,
                                              bos_token=special.get("bos_token", ""), eos_token=special.get("eos_token", ""),
                                              pad_token=special.get("pad_token", ""))

    def apply_chat_template(self, messages, tokenize: bool = False, add_generation_prompt: bool = True):
        if self.chat_template:
            text = self._render_template(messages, add_generation_prompt)
            return self.tokenizer.encode(text) if tokenize else text
        out = []
        for msg in messages:
            role, content = msg["role"], msg["content"]
            body = ""
            if isinstance(content, str):
                body = content
            else:
                for item in content:
                    if item.get("type") == "image" or "image" in item:
                        body += IMG_START + IMG_PAD + IMG_END
                    elif item.get("type") == "text":
                        body += item["text"]
            if role == "user":
                out.append(USER + body + END_USER)
            elif role == "assistant":
                out.append(ASSISTANT + body + END_ASSISTANT)
            else:                                   # system prompt: plain text prefix
                out.append(body)
        if add_generation_prompt:
            out.append(ASSISTANT)
        text = "".join(out)
        return self.tokenizer.encode(text) if tokenize else text

    # parser.py:99-105
    def __call__(self, text=None, images=None, videos=None, padding=True, return_tensors="pt", **_):
        import torch
        if isinstance(text, str):
            text = [text]
        images = list(images) if images is not None else []
        v = self.cfg.vision
        feats, grids, pv_dev = [], [], None
        if images and self.engine is not None and v.temporal_patch_size == 1:
            pv_dev, grids = self._preprocess_on_device(images)
        else:
            for im in images:
                pv, thw = preprocess_image(im, v.patch_size, v.spatial_merge_size, v.temporal_patch_size,
                                           self.cfg.min_pixels, self.cfg.max_pixels, self.cfg.image_mean, self.cfg.image_std)
                feats.append(pv)
                grids.append(thw)
        it = iter(grids)
        all_ids = []
        for t in text:
            pieces = t.split(IMG_PAD)
            expanded = pieces[0]
            for piece in pieces[1:]:
                thw = next(it, None)
                if thw is None:
                    raise ValueError("more image placeholders than images")
                expanded += IMG_PAD * (thw[0] * thw[1] * thw[2] // v.spatial_merge_size ** 2) + piece
            all_ids.append(self.tokenizer.encode(expanded))
        if next(it, None) is not None:
            raise ValueError("more images than image placeholders")
        L = max(len(x) for x in all_ids)
        pad = self.tokenizer.pad_token_id
        ids = np.full((len(all_ids), L), pad, dtype=np.int64)
        mask = np.zeros((len(all_ids), L), dtype=np.int64)
        for i, x in enumerate(all_ids):               # left padding: generation continues from the last column
            ids[i, L - len(x):] = x
            mask[i, L - len(x):] = 1
        data = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        if pv_dev is not None:
            data["pixel_values"] = pv_dev
            data["image_grid_thw"] = torch.tensor(grids, dtype=torch.int64)
        elif feats:
            data["pixel_values"] = torch.from_numpy(np.concatenate(feats, axis=0))
            data["image_grid_thw"] = torch.tensor(grids, dtype=torch.int64)
        return BatchFeature(data)

    # parser.py:114-116
    def batch_decode(self, sequences, skip_special_tokens: bool = True, clean_up_tokenization_spaces: bool = False):
        out = []
        for seq in sequences:
            ids = seq.tolist() if hasattr(seq, "tolist") else list(seq)
            out.append(self.tokenizer.decode(ids, skip_special_tokens=skip_special_tokens))
        return out

    def decode(self, ids, **kw):
        return self.batch_decode([ids], **kw)[0]
