"""Static vs continuous page batching on one MI355X, with page output lengths that differ.

Real pages stop at EOS after very different numbers of tokens; with random weights there is no meaningful EOS, so
every page gets its own length cap drawn from a seeded uniform distribution on [mean/4, 7*mean/4] (stated in the
output).  Static: consecutive batches of `slots` pages, each batch decoding until its longest page is done
(what dots_generate does).  Continuous: dots_ocr_amd.scheduler.ContinuousBatcher over the same slots.
Inputs (preprocessed pages) are resident in HBM before the clock starts, as in bench.py.

    python tools/serve_bench.py --pages 32 --slots 8 --mean-tokens 768
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=32)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--mean-tokens", type=int, default=768)
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--workload", default="a4", choices=["a4", "tiny"])
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.image_utils import preprocess_image
    from dots_ocr_amd.scheduler import ContinuousBatcher, Request
    from dots_ocr_amd.synthetic import A4_200DPI, synth_page, synth_prompt_ids
    from dots_ocr_amd.weights import random_state_dict

    if a.workload == "tiny":
        cfg, size = DotsConfig.tiny(layers=4, v_layers=4), (420, 588)
    else:
        cfg, size = DotsConfig(), A4_200DPI
    rng = np.random.default_rng(a.seed)
    caps = rng.integers(a.mean_tokens // 4, 7 * a.mean_tokens // 4 + 1, size=a.pages).astype(int).tolist()
    sd = random_state_dict(cfg, seed=a.seed, threads=min(32, os.cpu_count() or 8))

    distinct = min(a.pages, 8)
    feats, grids = zip(*(preprocess_image(synth_page(i, size)) for i in range(distinct)))
    dev = torch.device("cuda", 0)
    pv = [torch.from_numpy(f).to(dev) for f in feats]
    grid = [np.asarray([g], np.int64) for g in grids]
    prompts = [synth_prompt_ids(cfg, int(grids[i % distinct][1] * grids[i % distinct][2] // 4), seed=i) for i in range(a.pages)]
    plen = max(len(p) for p in prompts)
    eng = Engine(cfg, device=0, max_batch=a.slots, max_seq_len=plen + max(caps) + 64,
                 max_patches=a.slots * feats[0].shape[0] + 64, max_prefill_tokens=a.slots * plen + 64)
    eng.load_state_dict(sd)
    torch.cuda.synchronize()

    def run_static():
        steps, outs = 0, []
        for lo in range(0, a.pages, a.slots):
            sl = list(range(lo, min(a.pages, lo + a.slots)))
            pix = torch.cat([pv[i % distinct] for i in sl])
            torch.cuda.synchronize()
            out, _ = eng.generate(np.concatenate([prompts[i] for i in sl]), np.asarray([len(prompts[i]) for i in sl], np.int32),
                                  pix.data_ptr(), np.concatenate([grid[i % distinct] for i in sl]), max(caps[i] for i in sl), (),
                                  pixel_on_device=True)
            steps += max(caps[i] for i in sl) - 1
            outs += [out[j, :caps[i]] for j, i in enumerate(sl)]       # a page's tokens past its own stop are discarded
        return outs, steps

    def run_continuous(prefetch=0):
        cb = ContinuousBatcher(eng, chunk=a.chunk, prefetch=prefetch)
        outs = cb.run([Request(prompts[i], pv[i % distinct], grid[i % distinct], caps[i]) for i in range(a.pages)])
        return outs, cb.decode_steps

    res = {}
    # static batches are software-pipelined by the engine when called back to back?  No: run_static calls the plain generate (tower inside);
    # "continuous_prefetch1/2": the scheduler's look-ahead (the next requests' towers on the CU-masked side stream beside the running slots)
    modes = [("static", run_static), ("continuous", run_continuous), ("continuous_prefetch1", lambda: run_continuous(1)),
             ("continuous_prefetch2", lambda: run_continuous(2))]
    if a.slots >= 16:                                 # round 4: the partition plan covers any row count, so the look-ahead is measured at larger groups too
        modes += [("continuous_prefetch4", lambda: run_continuous(4)), ("continuous_prefetch8", lambda: run_continuous(8))]
    for name, fn in modes:
        fn() if a.workload == "tiny" else None                           # tiny: warm the graphs; full size: one cold run each
        eng.synchronize()
        t0 = time.perf_counter()
        outs, steps = fn()
        eng.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"seconds": round(dt, 3), "pages_per_s": round(a.pages / dt, 3), "tokens_per_s": round(sum(caps) / dt, 1),
                     "decode_steps": int(steps)}
        res[name + "_outs"] = outs
    outs = {k[:-5]: res.pop(k) for k in [k for k in res if k.endswith("_outs")]}
    ref = outs["continuous"]
    same = all(all(np.array_equal(x, y) for x, y in zip(ref, o)) for o in outs.values())
    print(json.dumps({"workload": a.workload, "pages": a.pages, "slots": a.slots, "chunk": a.chunk,
                      "length_caps": f"uniform[{a.mean_tokens // 4}, {7 * a.mean_tokens // 4}] seeded, sum {sum(caps)}",
                      "identical_tokens": bool(same), **res,
                      "speedup": round(res["static"]["seconds"] / res["continuous"]["seconds"], 3),
                      "speedup_prefetch1_vs_continuous": round(res["continuous"]["seconds"] / res["continuous_prefetch1"]["seconds"], 3)}))


if __name__ == "__main__":
    main()
