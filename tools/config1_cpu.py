#!/usr/bin/env python3
"""BASELINE.json configs[0] as BASELINE defines it, run ONCE on the build container's CPU: the reference's demo/demo_hf.py flow
(demo/demo_hf.py:10-51: apply_chat_template -> process_vision_info -> processor -> generate -> batch_decode, greedy,
max_new_tokens = 128) on the 1700 x 2250 demo page with the fp32 oracle (oracle/model.py: the PyTorch restatement of the HF path;
the reference's own model code is hub remote code that is not in /root/reference and flash-attn-only) — whole model,
all 42 + 28 layers, no extrapolation.  Random-init weights of the real architecture (no checkpoint offline).  Per-phase timers.

    python tools/config1_cpu.py [--threads 8] [--max-new-tokens 128] [--prompt-mode prompt_layout_all_en] [--out profiles/r03_config1_cpu.json]

Test infrastructure: this is the "plumbing, no GPU" configuration; nothing here is on the product path.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--max-new-tokens", type=int, default=128)
    ap.add_argument("--prompt-mode", default="prompt_layout_all_en")
    ap.add_argument("--image", default="/root/reference/demo/demo_image1.jpg")
    ap.add_argument("--out", default="profiles/r03_config1_cpu.json")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from dots_ocr.utils import dict_promptmode_to_prompt
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.processing import DotsOcrProcessor, process_vision_info
    from dots_ocr_amd.synthetic import synth_page
    from dots_ocr_amd.weights import random_state_dict
    from oracle import model as om

    T = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        out = fn()
        T[name] = time.perf_counter() - t0
        print(f"[config1] {name}: {T[name]:.2f} s", flush=True)
        return out

    cfg = DotsConfig()
    sd = timed("random_weights_fp32", lambda: {k: v.float() for k, v in random_state_dict(cfg, seed=0, threads=a.threads).items()})
    processor = DotsOcrProcessor(cfg)
    if Path(a.image).exists():
        image, image_desc = a.image, f"{a.image} (the reference's demo page)"
    else:
        image, image_desc = synth_page(1, (1700, 2250)), "synthetic 1700x2250 page (demo_image1.jpg's size)"
    prompt = dict_promptmode_to_prompt[a.prompt_mode]

    def host_pre():
        messages = [{"role": "user", "content": [{"type": "image", "image": image}, {"type": "text", "text": prompt}]}]
        text = processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
        image_inputs, video_inputs = process_vision_info(messages)
        return processor(text=[text], images=image_inputs, videos=video_inputs, padding=True, return_tensors="pt")

    inputs = timed("host_preprocess", host_pre)
    ids = inputs["input_ids"][0].to(torch.int64)
    pv = inputs["pixel_values"].float()
    grid = inputs["image_grid_thw"].to(torch.int64)
    with torch.no_grad():
        vis = timed("vision_tower", lambda: om.vision_tower(sd, cfg, pv, grid))
        emb = om.build_embeds(sd, cfg, ids, vis)
        cache = om.KVCache(cfg.num_hidden_layers)
        logits = timed("lm_prefill", lambda: om.lm_forward(sd, cfg, emb, cache))
        out = []

        def decode():
            nonlocal logits
            for step in range(a.max_new_tokens):
                tok = int(torch.argmax(logits[0]))
                out.append(tok)
                if step + 1 < a.max_new_tokens:
                    logits = om.lm_forward(sd, cfg, sd["model.embed_tokens.weight"][torch.tensor([tok])], cache)
        timed("decode", decode)
    text = timed("batch_decode", lambda: processor.batch_decode([np.asarray(out)], skip_special_tokens=True, clean_up_tokenization_spaces=False))
    total = T["host_preprocess"] + T["vision_tower"] + T["lm_prefill"] + T["decode"] + T["batch_decode"]
    rec = {
        "config": "BASELINE.json configs[0]: demo/demo_hf.py flow, CPU fp32, greedy, max_new_tokens=%d" % a.max_new_tokens,
        "what_ran": "oracle/model.py (fp32 restatement of the HF path), all %d vision blocks and %d LM layers, random-init weights" % (
            cfg.vision.num_hidden_layers, cfg.num_hidden_layers),
        "image": image_desc, "prompt_mode": a.prompt_mode, "threads": a.threads, "host": os.uname().nodename + " (build container)",
        "patches": int(pv.shape[0]), "vision_tokens": int(vis.shape[0]), "prompt_tokens": int(ids.shape[0]), "new_tokens": len(out),
        "seconds": {k: round(v, 3) for k, v in T.items()},
        "total_seconds_excl_weights": round(total, 2),
        "pages_per_s": round(1.0 / total, 6),
        "decode_tok_per_s": round((len(out) - 1) / T["decode"], 3),
        "output_text_head": text[0][:120],
        "first_tokens": out[:16],
    }
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
