"""Two engine instances, same weights and prompts: prefill + decode-step logits must be bitwise identical (GPU box helper)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dots_ocr_amd.config import DotsConfig  # noqa: E402
from dots_ocr_amd.engine import Engine  # noqa: E402
from dots_ocr_amd.weights import random_state_dict  # noqa: E402

cfg = DotsConfig.tiny(layers=3, v_layers=2, vocab=1024)
sd = random_state_dict(cfg, seed=3, threads=32)
rng = np.random.default_rng(0)
bad = 0
for B in (3, 11):
    lens = np.array([150 + 13 * i for i in range(B)], np.int32)
    ids = rng.integers(0, cfg.image_token_id, int(lens.sum())).astype(np.int32)
    runs = []
    for inst in range(3):
        e = Engine(cfg, max_batch=B, max_seq_len=640, max_patches=4096, max_prefill_tokens=B * 300)
        e.load_state_dict(sd)
        logits = []
        for rep in range(2):                      # twice on the same instance, too
            e.prefill(ids, lens)
            logits.append(e.get_logits().copy())
            for _ in range(6):
                e.decode_step()
                logits.append(e.get_logits().copy())
        runs.append(logits)
        e.close()
    ref = runs[0]
    for r, run in enumerate(runs):
        for i, (a, b) in enumerate(zip(ref[:7] * 2, run)):
            if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                bad += 1
                rows = np.unique(np.nonzero(a != b)[0]).tolist()
                print(f"B={B} instance {r} logits #{i}: differ in rows {rows}, max abs {np.abs(a - b).max():.5f}")
print("DETERMINISTIC" if not bad else f"NONDETERMINISTIC ({bad})")
