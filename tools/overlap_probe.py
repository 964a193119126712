#!/usr/bin/env python3
"""Does the ViT of the NEXT page batch overlap with the decode loop of the CURRENT one on this GPU?  Two engines (two HIP streams) in
one process: A runs the vision tower on 8 A4 pages, B decodes 8 sequences at the bench's context; timed alone and concurrently.
    python tools/overlap_probe.py [--steps 256]
"""
import argparse
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--split", type=int, default=0, help="CU-mask bits 0..split-1 for the decode engine, the rest for the ViT engine (0 = no masks)")
    a = ap.parse_args()
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.synthetic import synth_prompt_ids
    from dots_ocr_amd.weights import random_state_dict
    cfg = DotsConfig()
    sd = random_state_dict(cfg, seed=0, threads=32)
    N = 19824
    grid = np.asarray([[1, 118, 168]] * 8, np.int64)
    import os
    if a.split:
        os.environ["DOTS_OCR_CU_RANGE"] = f"{a.split}-255"
    A = Engine(cfg, max_batch=1, max_seq_len=256, max_patches=8 * N + 64, max_prefill_tokens=256)
    A.load_state_dict(sd)
    if a.split:
        os.environ["DOTS_OCR_CU_RANGE"] = f"0-{a.split - 1}"
    B = Engine(cfg, max_batch=8, max_seq_len=5200 + a.steps + 64, max_patches=256, max_prefill_tokens=8 * 5200 + 64)
    B.load_state_dict(sd)
    pix = torch.randn(8 * N, cfg.vision.patch_dim, device="cuda")
    torch.cuda.synchronize()
    ids = np.concatenate([synth_prompt_ids(cfg, 0, n_text_tokens=5200 - 3, seed=i)[:5200] for i in range(8)]).astype(np.int32)
    ids[ids == cfg.image_token_id] = 5
    lens = np.full(8, 5200, np.int32)

    def vit():
        t0 = time.perf_counter()
        A.vit_forward(pix.data_ptr(), grid, on_device=True)
        A.synchronize()
        return time.perf_counter() - t0

    def dec():
        t0 = time.perf_counter()
        B.generate(ids, lens, max_new_tokens=a.steps)
        dt = time.perf_counter() - t0
        return dt, B.stats()

    vit(); dec()
    tv = min(vit() for _ in range(2))
    td, st = dec()
    print(f"split {a.split}: alone: ViT {tv * 1e3:.0f} ms | prefill+decode({a.steps}) {td * 1e3:.0f} ms (decode {st['decode_ms']:.0f} ms, {st['decode_ms'] / max(1, st['decode_steps']):.3f} ms/step)")
    res = {}

    def run_v():
        res["v"] = vit()

    def run_d():
        res["d"] = dec()
    for rep in range(2):
        t0 = time.perf_counter()
        th = [threading.Thread(target=run_d), threading.Thread(target=run_v)]
        [t.start() for t in th]
        [t.join() for t in th]
        both = time.perf_counter() - t0
        print(f"concurrent: wall {both * 1e3:.0f} ms (sum alone {1e3 * (tv + td):.0f}); ViT took {res['v'] * 1e3:.0f} ms, prefill+decode took {res['d'][0] * 1e3:.0f} ms "
              f"(decode {res['d'][1]['decode_ms']:.0f} ms)")


if __name__ == "__main__":
    main()
