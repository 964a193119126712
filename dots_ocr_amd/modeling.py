"""``DotsOcrHipForCausalLM`` — the object DotsOCRParser installs as ``self.model``.

It honours the one model call the reference makes, ``self.model.generate(**inputs,
max_new_tokens=...) -> LongTensor[B, T+n]`` (dots_ocr/parser.py:110; demo/demo_hf.py:44), and the
``from_pretrained`` entry of parser.py:68-74, but everything underneath is the HIP engine: ViT,
merger, prefill and the hipGraph'd greedy decode loop run inside ``dots_generate`` on one GPU.

Greedy decoding only (``do_sample=False``): sampling at temperature is a SURVEY §8(f) "next" row.
There is no CPU fallback — constructing the model without the built library or without a GPU raises.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from .config import DotsConfig
from .engine import Engine
from .weights import load_state_dict, random_state_dict


def resolve_sampling(generation_config: dict, do_sample=None, temperature=None, top_p=None):
    """HF semantics: explicit generate() arguments win, otherwise the checkpoint's generation_config.json decides
    (GenerationMixin.generate merges it the same way; the reference calls generate(**inputs, max_new_tokens=...) only,
    parser.py:110).  Returns (temperature, top_p) for the engine: temperature 0 = greedy."""
    g = generation_config or {}
    do_sample = g.get("do_sample", False) if do_sample is None else do_sample
    temperature = g.get("temperature", 1.0) if temperature is None else temperature
    top_p = g.get("top_p", 1.0) if top_p is None else top_p
    if not do_sample or temperature is None or temperature <= 0:
        return 0.0, 1.0
    return float(temperature), float(min(max(top_p if top_p is not None else 1.0, 1e-6), 1.0))


def plan_batches(patches_per_seq: Sequence[int], max_batch: int, max_patches: int):
    """Consecutive sequences -> engine batches of <= max_batch sequences and <= max_patches vision patches (ViT workspace).
    A single sequence larger than the patch budget is rejected (it cannot be split: attention spans the whole image)."""
    batches, cur, used = [], [], 0
    for i, n in enumerate(patches_per_seq):
        if n > max_patches:
            raise ValueError(f"sequence {i} has {n} vision patches, more than the engine's max_patches={max_patches}")
        if cur and (len(cur) == max_batch or used + n > max_patches):
            batches.append(cur)
            cur, used = [], 0
        cur.append(i)
        used += n
    if cur:
        batches.append(cur)
    return batches


class DotsOcrHipForCausalLM:
    def __init__(self, cfg: DotsConfig, state_dict, device: int = 0, max_batch: int = 8, max_seq_len: int = 32768,
                 max_patches: Optional[int] = None, fp8_weights: bool = False):
        """fp8_weights: quantise the linears to e4m3 with per-output-channel scales at load time (DotsConfig.fp8_weights of the C ABI;
        DOTS_OCR_FP8=1 in the environment turns it on for callers that cannot pass the keyword, e.g. DotsOCRParser(use_hf=True))."""
        self.config = cfg
        self.device_index = device
        max_patches = max_patches or max(max_batch * 19824 + 64, 57600 + 64)
        fp8_weights = bool(fp8_weights) or os.environ.get("DOTS_OCR_FP8", "0") not in ("", "0")
        self.engine = Engine(cfg, device=device, max_batch=max_batch, max_seq_len=max_seq_len, max_patches=max_patches, fp8_weights=fp8_weights)
        self.engine.load_state_dict(state_dict)
        self.max_batch = max_batch
        self.max_seq_len = max_seq_len
        self.max_patches = max_patches
        self.generation_config = {"do_sample": False, "eos_token_id": list(cfg.eos_token_ids), "pad_token_id": cfg.pad_token_id}

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_pretrained(cls, model_path, device_map=None, torch_dtype=None, attn_implementation=None,
                        trust_remote_code=None, device: Optional[int] = None, **kw):
        """Accepts (and ignores) the HF keyword arguments the reference passes at parser.py:68-74."""
        model_path = Path(model_path)
        cfg = DotsConfig.from_pretrained(model_path)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        model = cls(cfg, load_state_dict(model_path), device=device, **kw)
        gen = model_path / "generation_config.json"
        if gen.exists():                                   # sampling defaults of the checkpoint (HF merges them into generate())
            import json
            g = json.loads(gen.read_text())
            for k in ("do_sample", "temperature", "top_p"):
                if k in g:
                    model.generation_config[k] = g[k]
        return model

    @classmethod
    def from_random(cls, cfg: Optional[DotsConfig] = None, seed: int = 0, device: int = 0, **kw):
        cfg = cfg or DotsConfig()
        return cls(cfg, random_state_dict(cfg, seed=seed), device=device, **kw)

    def eval(self):
        return self

    @property
    def device(self):
        import torch
        return torch.device("cuda", self.device_index)

    # ------------------------------------------------------------------ generate
    def generate(self, input_ids=None, attention_mask=None, pixel_values=None, image_grid_thw=None,
                 max_new_tokens: int = 128, do_sample: Optional[bool] = None, temperature: Optional[float] = None,
                 top_p: Optional[float] = None, seed: int = 0, eos_token_id=None, pad_token_id=None, continuous: Optional[bool] = None, **_):
        """HF-shaped generate.  do_sample / temperature / top_p default to the checkpoint's generation_config.json (greedy when it
        is absent); with sampling on, tokens are drawn on the GPU from softmax(logits / temperature) restricted to the top_p
        nucleus, reproducibly from `seed`.  Returns LongTensor
        [B, T + n]: the (padded) prompt followed by the new tokens, positions after a sequence's EOS filled with pad_token_id.

        More sequences than engine slots (B > max_batch) run with continuous batching: a slot is refilled with the next
        sequence as soon as its page hits EOS instead of waiting for the slowest page of a static batch
        (`continuous=True/False` forces either mode; greedy results are identical)."""
        import torch
        t_eff, p_eff = resolve_sampling(self.generation_config, do_sample, temperature, top_p)
        self.engine.set_sampling(t_eff, p_eff, seed if t_eff > 0 else 0)
        ids = input_ids.detach().cpu().numpy()
        B, T = ids.shape
        mask = attention_mask.detach().cpu().numpy().astype(bool) if attention_mask is not None else np.ones_like(ids, bool)
        eos = self.config.eos_token_ids if eos_token_id is None else eos_token_id
        eos = [eos] if isinstance(eos, int) else list(eos)
        pad = self.config.pad_token_id if pad_token_id is None else pad_token_id
        grid = image_grid_thw.detach().cpu().numpy().astype(np.int64) if image_grid_thw is not None else np.zeros((0, 3), np.int64)
        merge2 = self.config.vision.spatial_merge_size ** 2

        # which images belong to which sequence: image tokens are consumed in order
        prompts = [ids[b][mask[b]].astype(np.int32) for b in range(B)]
        n_img_tok = [int((p == self.config.image_token_id).sum()) for p in prompts]
        per_img_tok = (grid[:, 0] * grid[:, 1] * grid[:, 2] // merge2).tolist()
        img_of_seq, gi = [], 0
        for b in range(B):
            need, lst = n_img_tok[b], []
            while need > 0:
                if gi >= len(per_img_tok):
                    raise ValueError("image tokens do not match image_grid_thw")
                need -= per_img_tok[gi]
                lst.append(gi)
                gi += 1
            if need != 0:
                raise ValueError("image tokens do not match image_grid_thw")
            img_of_seq.append(lst)
        patch_off = np.concatenate([[0], np.cumsum(grid[:, 0] * grid[:, 1] * grid[:, 2])]).astype(np.int64)

        pv_dev, pv_host = None, None
        if pixel_values is not None:
            if pixel_values.is_cuda:
                pv_dev = pixel_values.contiguous().float()
                torch.cuda.synchronize(pv_dev.device)
            else:
                pv_host = np.ascontiguousarray(pixel_values.detach().numpy(), dtype=np.float32)

        # like HF, generation stops at the context capacity instead of failing: the reference asks for 24 000 new tokens
        # (parser.py:110) on top of prompts of up to 14 400 vision tokens; the KV pool is sized for max_seq_len per sequence
        longest = max(len(p) for p in prompts)
        if longest >= self.max_seq_len:
            raise ValueError(f"prompt of {longest} tokens does not fit max_seq_len={self.max_seq_len}")
        max_new_tokens = max(1, min(int(max_new_tokens), self.max_seq_len - longest))
        new_tokens = np.full((B, max_new_tokens), pad, dtype=np.int64)
        n_max = 0
        seq_patches = [int(sum(patch_off[g + 1] - patch_off[g] for g in img_of_seq[b])) for b in range(B)]
        if continuous is None:
            continuous = B > self.max_batch
        if continuous:
            from .scheduler import ContinuousBatcher, Request
            reqs = []
            for b in range(B):
                if img_of_seq[b]:
                    lo, hi = int(patch_off[img_of_seq[b][0]]), int(patch_off[img_of_seq[b][-1] + 1])
                    pix = pv_dev[lo:hi] if pv_dev is not None else pv_host[lo:hi]
                    reqs.append(Request(prompts[b], pix, grid[img_of_seq[b][0]:img_of_seq[b][-1] + 1], max_new_tokens))
                else:
                    reqs.append(Request(prompts[b], None, None, max_new_tokens))
            outs = ContinuousBatcher(self.engine, eos_ids=eos).run(reqs)
            for b, o in enumerate(outs):
                new_tokens[b, :len(o)] = o
                n_max = max(n_max, len(o))
        # static batches within the engine's capacity.  With several batches the vision tower of batch k+1 is prefetched on the engine's
        # CU-masked side stream while batch k is prefilled and decoded (Engine.vit_prefetch: same tokens, ~20 % more pages/s at 8 x A4).
        plan = [] if continuous else plan_batches(seq_patches, self.max_batch, self.max_patches)

        def pixels_of(sl):
            imgs = [g for b in sl for g in img_of_seq[b]]
            if not imgs:
                return None, None, False
            lo, hi = int(patch_off[imgs[0]]), int(patch_off[imgs[-1] + 1])         # images of a slice are contiguous
            g = grid[imgs[0]:imgs[-1] + 1]
            if pv_dev is not None:
                return pv_dev.data_ptr() + lo * pv_dev.shape[1] * 4, g, True
            return pv_host[lo:hi], g, False
        pipelined = len(plan) > 1 and all(pixels_of(sl)[0] is not None for sl in plan) and hasattr(self.engine, "vit_prefetch")
        if pipelined:
            pix, g, on_dev = pixels_of(plan[0])
            self.engine.vit_prefetch(pix, g, on_device=on_dev)
        prefetched = pipelined                   # a tower is in flight / waiting to be taken
        try:
            for k, sl in enumerate(plan):
                lens = np.array([len(prompts[b]) for b in sl], np.int32)
                packed = np.concatenate([prompts[b] for b in sl])
                pix, g, on_dev = pixels_of(sl)
                if pipelined:
                    self.engine.vit_take()
                    prefetched = False
                    if k + 1 < len(plan):
                        npix, ng, non_dev = pixels_of(plan[k + 1])
                        self.engine.vit_prefetch(npix, ng, on_device=non_dev, after_prefill=True)
                        prefetched = True
                    out, out_lens = self.engine.generate(packed, lens, max_new_tokens=max_new_tokens, eos_ids=eos, vision_taken=True)
                elif pix is not None:
                    out, out_lens = self.engine.generate(packed, lens, pix, g, max_new_tokens, eos, on_dev)
                else:
                    out, out_lens = self.engine.generate(packed, lens, None, None, max_new_tokens, eos)
                for j, b in enumerate(sl):
                    new_tokens[b, :out_lens[j]] = out[j, :out_lens[j]]
                    n_max = max(n_max, int(out_lens[j]))
        except Exception:
            if prefetched:                           # leave the engine usable: a waiting prefetch would refuse the next one
                try:
                    self.engine.vit_take()
                except Exception:
                    pass
            raise
        full = np.concatenate([ids.astype(np.int64), new_tokens[:, :n_max]], axis=1)    # HF stops at the longest sequence
        res = torch.from_numpy(full)
        return res.to(input_ids.device) if input_ids.is_cuda else res

    def stats(self) -> dict:
        return self.engine.stats()
