#!/usr/bin/env python3
"""Anchor for bench.py's extrapolated `cpu_baseline`: ONE full-depth synthetic A4 page through the fp32 CPU
oracle — all 42 ViT blocks, all 28 LM layers of the 5 200-token prefill, N full-depth decode steps — timed on this machine's cores, next
to bench.py's own 1-vs-3-layer extrapolation run on the SAME machine with the SAME thread count.  Offline (tens of minutes); writes
profiles/r05_cpu_anchor.json.   usage: python tools/cpu_anchor.py [threads] [decode_steps]"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from dots_ocr_amd.config import DotsConfig  # noqa: E402
from dots_ocr_amd.image_utils import preprocess_image  # noqa: E402
from dots_ocr_amd.synthetic import A4_200DPI, synth_page, synth_prompt_ids  # noqa: E402
from dots_ocr_amd.weights import random_state_dict  # noqa: E402
from oracle import model as om  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
n_dec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
NEW = 1024
torch.set_num_threads(threads)
cfg = DotsConfig()
sd = random_state_dict(cfg, seed=0, threads=min(32, threads))
page = synth_page(0, A4_200DPI)
ids = synth_prompt_ids(cfg, 19824 // 4, seed=0)
rec = {"machine": {"cpu_count": os.cpu_count(), "threads": threads}, "page": "synthetic A4@200dpi: 19824 patches, %d prompt tokens, %d new tokens" % (len(ids), NEW)}
t0 = time.perf_counter()
rec["extrapolated"] = bench.cpu_baseline(cfg, sd, threads, page, ids, NEW)
rec["extrapolation_wall_s"] = time.perf_counter() - t0
print("extrapolated:", rec["extrapolated"]["value"], "pages/s", flush=True)
sdf = {k: v.float() for k, v in sd.items()}
del sd
pv, thw = preprocess_image(page)
with torch.no_grad():
    t0 = time.perf_counter()
    vis = om.vision_tower(sdf, cfg, torch.from_numpy(pv), torch.tensor([thw]))
    t_vit = time.perf_counter() - t0
    print("full-depth tower: %.1f s" % t_vit, flush=True)
    emb = om.build_embeds(sdf, cfg, torch.from_numpy(ids.astype(np.int64)), vis)
    cache = om.KVCache(cfg.num_hidden_layers)
    t0 = time.perf_counter()
    logits = om.lm_forward(sdf, cfg, emb, cache)
    t_pre = time.perf_counter() - t0
    print("full-depth prefill: %.1f s" % t_pre, flush=True)
    tok = torch.tensor([int(torch.argmax(logits[0]))])
    t0 = time.perf_counter()
    for _ in range(n_dec):
        logits = om.lm_forward(sdf, cfg, sdf["model.embed_tokens.weight"][tok], cache)
        tok = torch.tensor([int(torch.argmax(logits[0]))])
    t_step = (time.perf_counter() - t0) / n_dec
t_page = t_vit + t_pre + t_step * (NEW - 1)
rec["full_depth"] = {"vit_42_blocks_s": t_vit, "prefill_28_layers_s": t_pre, "decode_step_28_layers_s": t_step, "decode_steps_timed": n_dec,
                     "page_s": t_page, "pages_per_s": 1.0 / t_page,
                     "note": "tower and prefill measured in full; the %d decode steps (every step costs the same at fixed context) scaled to %d" % (n_dec, NEW - 1)}
ex = rec["extrapolated"]["value"]
rec["extrapolated_over_measured"] = ex / rec["full_depth"]["pages_per_s"]
out = ROOT / "profiles" / "r05_cpu_anchor.json"
out.write_text(json.dumps(rec, indent=1))
print(json.dumps({k: v for k, v in rec.items() if k != "extrapolated"}, indent=1))
