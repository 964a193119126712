"""Persistent decode step vs launch-per-phase step on the same weights and prompts: bitwise logits and token equality,
then step time.  GPU box helper; exits non-zero on mismatch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dots_ocr_amd.config import DotsConfig  # noqa: E402
from dots_ocr_amd.engine import Engine  # noqa: E402
from dots_ocr_amd.weights import random_state_dict  # noqa: E402

full = len(sys.argv) > 1 and sys.argv[1] == "full"
cfg = DotsConfig() if full else DotsConfig.tiny(layers=3, v_layers=2, vocab=1024)
sd = random_state_dict(cfg, seed=3, threads=32)
rng = np.random.default_rng(0)


def make(persistent, max_batch):
    os.environ["DOTS_OCR_PERSISTENT"] = "1" if persistent else "0"
    e = Engine(cfg, max_batch=max_batch, max_seq_len=6144 if full else 640, max_patches=4096, max_prefill_tokens=max_batch * (5200 if full else 300))
    e.load_state_dict(sd)
    return e


ok = True
for B in ((8,) if full else (3, 11)):
    lens = np.array([(5000 if full else 150) + 13 * i for i in range(B)], np.int32)
    ids = rng.integers(0, min(cfg.vocab_size - 8, cfg.image_token_id), int(lens.sum())).astype(np.int32)
    res = {}
    for persistent in (False, True):
        e = make(persistent, B)
        e.prefill(ids, lens)
        logits = []
        for _ in range(4):
            e.decode_step()
            logits.append(e.get_logits().copy())
        n_new = 64 if full else 40
        out, n = e.generate(ids, lens, None, None, n_new, ())
        e.synchronize()
        t0 = time.perf_counter()
        out2, _ = e.generate(ids, lens, None, None, n_new, ())
        e.synchronize()
        dt = time.perf_counter() - t0
        st = e.stats()
        res[persistent] = (logits, out, st["decode_ms"] / max(1, st["decode_steps"]))
        e.close()
    for i, (a, b) in enumerate(zip(res[False][0], res[True][0])):
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if not same:
            ok = False
            print(f"B={B} step {i}: logits differ, max abs {np.abs(a - b).max():.5f}, nan {np.isnan(b).sum()}")
    tok_same = np.array_equal(res[False][1], res[True][1])
    ok = ok and tok_same
    print(f"B={B}: tokens equal {tok_same}; ms/step launch-per-phase {res[False][2]:.4f}  persistent {res[True][2]:.4f}", flush=True)
print("PERSIST_OK" if ok else "PERSIST_MISMATCH")
sys.exit(0 if ok else 1)
