#!/usr/bin/env python3
"""Headline benchmark: pages/sec + output tok/s, dots.ocr 1.7B bf16, A4@200dpi page batch
(BASELINE.json metric; default workload = configs[1]: batch of 8 synthetic A4 pages per GPU, layout-all
prompt, greedy, max_new_tokens=1024 with EOS disabled so every page emits exactly 1024 tokens).

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path (reference dots_ocr/parser.py:78-117) over one batch, END TO END from the
uint8 page pixels resident in HBM to strings on the host:
    chat template + tokenisation (host)  ->  bicubic resize / normalise / patchify (GPU, Pillow-exact)  ->  ViT + merger
    ->  LM prefill  ->  hipGraph'd greedy decode loop  ->  token ids to the host  ->  detokenisation.
Data-parallel: one process per GPU, a full replica each, pages sharded by page, the only collective being the final
gather of token ids (RCCL).  --workload a4 / highres: weak scaling (B pages per GPU).  --workload mixed64: BASELINE
configs[3], 64 mixed-size pages for the whole job, cost-sharded (LPT) over the ranks, continuous batching per rank: strong scaling.
--workload svg: BASELINE configs[4] — one 588x560 chart page (420 vision tokens), prompt_image_to_svg with {width}/{height}
filled (reference demo/demo_vllm_svg.py:28), 4096 new tokens, GREEDY (the reference samples this task at T = 0.9), fp8 (e4m3,
per-output-channel scale) weights; B = 1.  --fp8 0/1 overrides the weight format of any workload.

Prints ONE JSON line on rank 0 (see README/DESIGN for the field contract).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

_USER_SET_DEC_CUS = "DOTS_OCR_OVERLAP_DEC_CUS" in os.environ      # (before main() picks the workload's default partition)
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak (same guide; ~6.3 TB/s achievable)
SVG_CHART = (588, 560)         # demo/demo_image2.png-sized chart input
N_TEXT_TOKENS = 241            # BPE length of the chat template around prompt_layout_all_en (T = 4956 + ~244)

# BASELINE configs[3] / SURVEY §8(d) config 4: empirical fixture distribution of page sizes (width, height), seed 2025
MIXED_SIZES = [((1654, 2339), 0.50), ((1700, 2250), 0.15), ((2339, 3308), 0.10), ((1344, 1344), 0.10), ((946, 1024), 0.10), ((583, 550), 0.05)]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="pages per GPU per step (mixed64: sequence slots per GPU, default 32)")
    ap.add_argument("--max-new-tokens", type=int, default=1024)
    ap.add_argument("--workload", default="a4", choices=["a4", "highres", "tiny", "mixed64", "svg"])
    ap.add_argument("--fp8", type=int, default=None, help="1: e4m3 per-channel weights (DotsConfig.fp8_weights); default: on for --workload svg only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="strictly sequential batches (ViT -> prefill -> decode); default: the tower of batch k+1 runs on a CU-masked side stream "
                         "while batch k decodes on the complementary CU partition (dots_vit_prefetch)")
    ap.add_argument("--rows-in-flight", type=int, default=None,
                    help="a4 / highres with overlap: decode rows in flight = how many page batches decode together (continuous batching over "
                         "sequence slots, admission in groups of --batch pages).  Default: the engine's 64 sequence slots (8 batches of 8 pages): the "
                         "last rows-in-flight / batch batches share every decode step (each weight byte is read once per 64 rows instead of once per 8) "
                         "while the tower of the next batch runs on its CU partition; per timed step still exactly one preprocessing pass, one tower, one "
                         "prefill and batch x max_new_tokens decoded tokens.  = batch: the round-3 pipeline (one batch decoding)")
    ap.add_argument("--time-sliced", action="store_true",
                    help="the multi-batch pipeline WITHOUT CU partitions: per step the tower of the new batch, its prefill and the decode steps over all "
                         "rows in flight run one after the other, each on the whole chip (A/B against the partitioned default)")
    ap.add_argument("--page-queue", action="store_true",
                    help="mixed64: the ranks PULL pages from one shared queue (dp.PageQueue: a counter on the process group's host-side store, costliest pages "
                         "first) as their slots drain, instead of the static cost shard — for jobs much larger than the ranks' slots, where a static shard "
                         "leaves the ranks that drew short outputs idle at the end (SURVEY §8(e)).  Every rank keeps all pages' pixels resident.")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default a4 run on one GPU: skip the short highres / mixed64 / svg-fp8 legs whose lines are embedded under `other_configs`")
    ap.add_argument("--page-sets", type=int, default=None,
                    help="a4 / highres / svg: distinct page sets the steps rotate through (default 2 with the pipelined step, else 1): step k processes set k mod n, "
                         "all sets resident in HBM, each set's tokens checked against its own sequential batch")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def respawn_under_torchrun(a):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N ranks (one per GPU) through torch.distributed.run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    os.execvp(cmd[0], cmd)


def cpu_baseline(cfg, sd_bf16, cores, page, ids, max_new_tokens):
    """The fp32 CPU oracle (PyTorch restatement of the reference's HF path, oracle/model.py) timed on a BOUNDED sample of the
    SAME unit of work — one full A4 page with its 5 200-token prompt — and extrapolated exactly by layer count: every
    ViT block costs the same, every LM layer costs the same, every decode step (at fixed context) costs the same.  Timed with a
    FIXED thread count: the tower with 1 and with 3 blocks, the LM prefill with 1 and with 3 layers, 4 decode steps with 1 and
    with 3 layers (per-layer cost = half the difference: two layers' worth of signal instead of the 0-vs-1 difference of round 2);
    All raw timings are in the record (the tower pair timed twice differed by 0.9 % in extrapolation: profiles/r03_bench_a4.json).
    Checker code used only as the baseline being reported, never on the product path."""
    import copy
    import torch
    from dots_ocr_amd.image_utils import preprocess_image
    from oracle import model as om
    torch.set_num_threads(cores)
    pv, thw = preprocess_image(page)
    t_ids = torch.from_numpy(ids.astype(np.int64))
    grid = torch.tensor([thw])
    first3 = lambda name, pre: any(name.startswith(f"{pre}{i}.") for i in range(3))
    keep = lambda name: (not name.startswith("vision_tower.blocks.") or first3(name, "vision_tower.blocks.")) and \
        (not name.startswith("model.layers.") or first3(name, "model.layers."))
    sd = {k: v.float() for k, v in sd_bf16.items() if keep(k)}

    def cfg_with(v_layers, layers):
        c = copy.deepcopy(cfg)
        c.vision.num_hidden_layers, c.num_hidden_layers = v_layers, layers
        return c

    def timed(fn):
        t0 = time.perf_counter()
        out = fn()
        return time.perf_counter() - t0, out
    raw = {"vit_1_block_s": [], "vit_3_blocks_s": []}
    with torch.no_grad():
        om.vision_tower(sd, cfg_with(1, 0), torch.from_numpy(pv[:2048]), torch.tensor([[1, 32, 64]]))       # thread pool / allocator warm-up
        vis = None
        for _ in range(1):                           # (a second pass differed by 0.9 % in the round-3 run: profiles/r03_bench_a4.json; one pass keeps the leg at ~70 s)
            t1, vis = timed(lambda: om.vision_tower(sd, cfg_with(1, 0), torch.from_numpy(pv), grid))
            t3, _ = timed(lambda: om.vision_tower(sd, cfg_with(3, 0), torch.from_numpy(pv), grid))
            raw["vit_1_block_s"].append(t1)
            raw["vit_3_blocks_s"].append(t3)
        emb = om.build_embeds(sd, cfg, t_ids, vis)
        n_dec = 16
        res = {}
        for layers in (1, 3):
            c = cfg_with(1, layers)
            cache = om.KVCache(layers)
            t_p, logits = timed(lambda: om.lm_forward(sd, c, emb, cache))
            tok = torch.tensor([int(torch.argmax(logits[0]))])
            t_d, _ = timed(lambda: [om.lm_forward(sd, c, sd["model.embed_tokens.weight"][tok], cache) for _ in range(n_dec)])
            res[layers] = (t_p, t_d / n_dec)
            raw[f"prefill_{layers}_layers_s"], raw[f"decode_step_{layers}_layers_s"] = t_p, t_d / n_dec
    VL, LL = cfg.vision.num_hidden_layers, cfg.num_hidden_layers

    def extrapolate(t_one, t_three, n):                      # t(n) = t_one + (n - 1) * per_layer
        return t_one + (n - 1) * (t_three - t_one) / 2.0
    vit_runs = [extrapolate(a, b, VL) for a, b in zip(raw["vit_1_block_s"], raw["vit_3_blocks_s"])]
    t_vit = sum(vit_runs) / len(vit_runs)
    t_prefill = extrapolate(res[1][0], res[3][0], LL)
    t_step = extrapolate(res[1][1], res[3][1], LL)
    t_page = t_vit + t_prefill + t_step * (max_new_tokens - 1)
    measured = sum(raw["vit_1_block_s"]) + sum(raw["vit_3_blocks_s"]) + sum(r[0] + r[1] * n_dec for r in res.values())
    spread = (max(vit_runs) - min(vit_runs)) / t_vit
    return {"value": 1.0 / t_page, "unit": "pages/s", "cores": cores, "kind": "port",
            "sample": f"fp32 oracle on ONE full synthetic A4 page ({pv.shape[0]} patches, {len(ids)} prompt tokens, {max_new_tokens} new tokens), "
                      f"{measured:.1f} s measured with {cores} threads: ViT with 1 and 3 of {VL} blocks" + (f" ({len(vit_runs)} passes; their extrapolations differ by {spread * 100:.1f} %)" if len(vit_runs) > 1 else "") + f", LM prefill and {n_dec} decode steps with 1 and 3 of {LL} layers; extrapolated by layer count to "
                      f"{t_page:.0f} s per page (ViT {t_vit:.0f} s, prefill {t_prefill:.0f} s, decode {t_step * 1e3:.0f} ms/token = {1.0 / t_step:.2f} tok/s)",
            "raw_seconds": {k: ([round(x, 3) for x in v] if isinstance(v, list) else round(v, 4)) for k, v in raw.items()}}


def bench_prompt_ids(proc, cfg, messages, n_vis, page_no):
    """chat template -> ids.  The stand-in byte tokenizer (no checkpoint tokenizer exists offline) makes ~4x more
    tokens than BPE for the text part, so the text is clipped to the BPE-equivalent length: the prompt keeps the
    BASELINE shape (3 + n_vis + 241 + ... = 5 200 tokens for an A4 page) while its tokenisation is really executed.
    (Module level so that tools/make_a4_anchor.py and tests/test_a4_anchor_gpu.py build the very prompt the bench runs.)"""
    from dots_ocr_amd.processing import IMG_PAD
    text = proc.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    head, tail = text.split(IMG_PAD)                           # "<|user|><|img|>" | "<|endofimg|>{prompt}<|endofuser|><|assistant|>"
    h_ids, t_ids = proc.tokenizer.encode(head), proc.tokenizer.encode(tail)
    rng = np.random.default_rng(page_no)                       # distinct prompts per page, like distinct documents
    body = np.asarray(t_ids[1:-2][:N_TEXT_TOKENS - 2], np.int64)
    body = (body + rng.integers(0, 200, len(body))) % 256      # still byte ids
    return np.concatenate([h_ids, [cfg.image_token_id] * n_vis, t_ids[:1], body, t_ids[-2:]]).astype(np.int32)


def bench_messages(workload="a4"):
    prompts_json = json.loads((ROOT / "dots_ocr_amd" / "data" / "prompts.json").read_text())
    if workload == "svg":
        prompt_text = prompts_json["prompt_image_to_svg"].replace("{width}", str(SVG_CHART[0])).replace("{height}", str(SVG_CHART[1]))
    else:
        prompt_text = prompts_json["prompt_layout_all_en"]
    return [{"role": "user", "content": [{"type": "image", "image": "page"}, {"type": "text", "text": prompt_text}]}]


def other_config_legs():
    """BASELINE configs[2] (highres), [3] (mixed64 on this one GPU) and [4] (svg, fp8) as short legs of the DEFAULT run, so that the driver's own record
    holds them (VERDICT r5 missing #3: they existed only as builder-run files under profiles/).  Each leg is this script in a fresh process (its own
    engine, its own parity check — a leg that is not parity-clean exits non-zero and is reported as failed); the headline keys of the a4 line are
    untouched.  About 60 s in all."""
    import subprocess
    legs = [("highres", ["--workload", "highres", "--batch", "4", "--steps", "8", "--warmup", "4"]),      # configs[2]: batch = 4      # (the adaptive tower tail settles within the warm-up steps: 3 / 1 measured 11.1 pages/s against 14.4 at 5 / 2)
            ("mixed64", ["--workload", "mixed64", "--steps", "1", "--warmup", "0"]),
            ("svg_fp8", ["--workload", "svg", "--steps", "1", "--warmup", "1"])]
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "output_tok_s", "roofline", "roofline_decode", "roofline_decode_sequential", "roofline_decode_alone_rows_in_flight",
            "parity_vs_sequential", "steps_checked", "page_sets", "parity_vs_single_sequence", "pages_checked", "phase_ms_per_step", "h2d", "setup_s")
    out = {}
    for name, argv in legs:
        t0 = time.perf_counter()
        try:
            # a clean environment for the leg: the decode partition this process chose for ITS workload (setdefault below) must not become the leg's
            # (round 6: the highres leg inherited a4's 64 decode CUs and reported 11.4 pages/s instead of its own default's 13.8)
            env = {k: v for k, v in os.environ.items() if not (k == "DOTS_OCR_OVERLAP_DEC_CUS" and not _USER_SET_DEC_CUS)}
            p = subprocess.run([sys.executable, str(Path(__file__).resolve()), *argv, "--no-cpu-baseline", "--no-other-configs"], capture_output=True, text=True, timeout=420, env=env)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
            if p.returncode != 0 or not line:
                out[name] = {"failed": True, "rc": p.returncode, "stderr_tail": p.stderr[-600:]}
            else:
                d = json.loads(line[-1])
                out[name] = {k: d[k] for k in keep if k in d}
                out[name]["workload"] = d.get("config", {}).get("workload")
        except Exception as e:                       # a leg must never take the headline line down with it
            out[name] = {"failed": True, "error": repr(e)[:300]}
        out[name]["leg_wall_s"] = time.perf_counter() - t0
    return out


def mixed_pages(n_total, seed=2025):
    import random
    rng = random.Random(seed)
    sizes, weights = zip(*MIXED_SIZES)
    return [rng.choices(sizes, weights)[0] for _ in range(n_total)]


def main():
    a = parse()
    if a.workload == "svg":                         # BASELINE configs[4]: one chart page, chart->SVG prompt, 4096 new tokens, fp8 weights
        argv = " ".join(sys.argv[1:])
        if "--batch" not in argv:
            a.batch = 1
        if "--max-new-tokens" not in argv:
            a.max_new_tokens = 4096
    if a.workload == "mixed64" and "--batch" not in " ".join(sys.argv[1:]):
        a.batch = 32                                # sequence slots per rank (engine maximum 64): decode is latency-bound, extra rows are cheap.
                                                    # 1 GPU: 2.90 / 3.40 / 3.92 / 4.06 pages/s with 8 / 16 / 32 / 64 slots
    fp8 = bool(a.fp8) if a.fp8 is not None else a.workload == "svg"
    if a.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(a)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the dots.ocr HIP engine has no CPU fallback")
    torch.cuda.set_device(local)
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # launched by torch.distributed.run (any world size)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from dots_ocr_amd import dp
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.image_utils import smart_resize
    from dots_ocr_amd.processing import IMG_PAD, DotsOcrProcessor
    from dots_ocr_amd.synthetic import A4_200DPI, HIGH_RES, synth_page
    from dots_ocr_amd.weights import random_state_dict

    cfg = DotsConfig.tiny(layers=4, v_layers=4) if a.workload == "tiny" else DotsConfig()
    mixed = a.workload == "mixed64"
    B = a.batch
    t_setup = time.perf_counter()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    sd = random_state_dict(cfg, seed=a.seed, threads=max(2, min(32, (os.cpu_count() or 8) // max(1, local_world))))    # N ranks share the host's cores
    v = cfg.vision
    factor = v.patch_size * v.spatial_merge_size

    def patches_of(size):
        rh, rw = smart_resize(size[1], size[0], factor, cfg.min_pixels, cfg.max_pixels)
        return (rh // v.patch_size) * (rw // v.patch_size)

    # ---- this rank's pages
    if mixed:
        sizes_all = mixed_pages(64)
        costs_all = [dp.page_cost(patches_of(sz), a.max_new_tokens) for sz in sizes_all]
        shards = dp.shard_pages(costs_all, world)
        my_pages = list(range(len(sizes_all))) if a.page_queue else shards[rank]      # --page-queue: any page may come to this rank, all are kept resident
        n_job_pages = len(sizes_all)
    else:
        size = {"tiny": (420, 588), "a4": A4_200DPI, "highres": HIGH_RES, "svg": SVG_CHART}[a.workload]
        sizes_all = None
        my_pages = [rank * B + i for i in range(B)]
        n_job_pages = world * B
    sizes = [sizes_all[i] if mixed else size for i in my_pages]
    # the steps of the timed region rotate through n_sets DISTINCT page sets (every step used to re-process the same 8 resident pages;
    # nothing was cached, but now nothing can be): set s of this rank = pages (s * world + rank) * B ... + B - 1, same size, own text and own prompts
    n_sets = 1 if mixed else max(1, a.page_sets if a.page_sets is not None else (1 if a.no_overlap else 2))
    sets = [my_pages] if mixed else [[(s_ * world + rank) * B + i for i in range(B)] for s_ in range(n_sets)]
    set_pages = [[synth_page(i, sz) for i, sz in zip(pg, sizes)] for pg in sets]
    pages = set_pages[0]
    n_patches = [patches_of(sz) for sz in sizes]
    max_prompt = max(n_patches) // 4 + N_TEXT_TOKENS + 3
    max_seq = max_prompt + a.max_new_tokens + 64
    slots = B
    overlap = not a.no_overlap and not mixed
    rif = a.rows_in_flight if a.rows_in_flight is not None else (max(B, 64 // B * B) if overlap and a.workload in ("a4", "highres") and B <= 64 else B)
    deep = overlap and rif > B                      # several page batches decode together (sequence slots, admission in groups of B)
    sliced = deep and a.time_sliced                 # same admission pattern, no CU partitions: tower, prefill and decode steps take turns on the whole chip
    if deep and (rif % B or rif > 64):
        raise SystemExit("--rows-in-flight must be a multiple of --batch, at most 64 (the engine's sequence slots)")
    n_groups = rif // B if deep else 1
    if mixed:
        # look-ahead towers on 192 CUs beside up to 32 decoding rows on 64 (round 5, with the wide decode kernels: 4.92 pages/s against 4.68 at
        # 96 / 160, profiles/r05_partition_sizes_highres_mixed64.txt; round 4: 96)
        os.environ.setdefault("DOTS_OCR_OVERLAP_DEC_CUS", "64")
    if deep and not sliced:
        # measured (profiles/r04_deep_sweep.txt; decode CUs / rows in flight): a4 64 / 64 5.12 pages/s (tower 1 358 ms on 192 CUs, decode 1 418 ms; 5.34-5.43 with two batch tiles per workgroup in gate|up / lm_head: decode 1 270 ms), 64 / 32 4.93,
        # 96 / 32 4.95, 96 / 64 4.93, one batch on 128 / 128 4.11-4.21; highres 128 / 64 10.4 (11.5 with the two-tile kernels), 128 / 32 9.0, 128 / 16 7.3, 96 / 32 8.1
        os.environ.setdefault("DOTS_OCR_OVERLAP_DEC_CUS", "64" if a.workload == "a4" else "128")
    max_patches = max(sum(sorted(n_patches, reverse=True)[:slots]), max(n_patches)) + 64
    eng = Engine(cfg, device=local, max_batch=rif if deep else slots, max_seq_len=max_seq, max_patches=max_patches,
                 max_prefill_tokens=slots * max_prompt + 64, fp8_weights=fp8)
    eng.load_state_dict(sd)
    proc = DotsOcrProcessor(cfg, engine=eng)
    messages = bench_messages(a.workload)

    # inputs resident in HBM before the timed region: the uint8 pixels of every page; one fp32 patch buffer is reused
    set_arrays = [[np.ascontiguousarray(np.asarray(p.convert("RGB"), dtype=np.uint8)) for p in pg] for pg in set_pages]
    set_dev = [[eng.to_device(arr) for arr in arrs] for arrs in set_arrays]
    page_arrays, page_dev = set_arrays[0], set_dev[0]
    pix = torch.empty((sum(n_patches), v.patch_dim), dtype=torch.float32, device=torch.device("cuda", local))
    pix_dev = pix.data_ptr()
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    host_ms = {"tokenize_ms": 0.0, "preprocess_ms": 0.0, "detokenize_ms": 0.0}

    def tokenize(n_vis, page_no):
        return bench_prompt_ids(proc, cfg, messages, n_vis, page_no)

    def prompts_of(sid):
        return [tokenize(n // 4, pn) for n, pn in zip(n_patches, sets[sid])]

    def preprocess_all(sid=0):
        grids, off = [], 0
        for dptr, arr, n in zip(set_dev[sid], set_arrays[sid], n_patches):
            grids.append(eng.preprocess_image(dptr, pix_dev + off * v.patch_dim * 4, shape=arr.shape[:2]))
            off += n
        return grids

    dyn = {"job": 0, "pages": None}

    def step_page_queue(t0):
        """mixed64 with --page-queue: this rank pulls page indices from the job's shared queue as its slots drain; a pulled page is tokenised and
        preprocessed (GPU) then, so per job every page is still preprocessed, encoded, prefilled and decoded exactly once over all ranks."""
        from dots_ocr_amd.scheduler import ContinuousBatcher, Request
        pq = dp.PageQueue(costs_all, store=dp.PageQueue.default_store(), key="dots_ocr/page_queue/%d" % dyn["job"], world_size=world)
        dyn["job"] += 1
        offs = np.concatenate([[0], np.cumsum(n_patches)])
        t_host = [0.0, 0.0]

        def pull(k):
            got = []
            for p_ in pq.take(k):
                ta = time.perf_counter()
                pr = tokenize(n_patches[p_] // 4, p_)
                tb = time.perf_counter()
                g = eng.preprocess_image(page_dev[p_], pix_dev + int(offs[p_]) * v.patch_dim * 4, shape=page_arrays[p_].shape[:2])
                t_host[0] += tb - ta; t_host[1] += time.perf_counter() - tb
                got.append((p_, Request(pr, pix[int(offs[p_]):int(offs[p_ + 1])], np.asarray([g], np.int64), a.max_new_tokens)))
            return got
        ag = os.environ.get("DOTS_BENCH_ADMIT_GROUP")
        cb = ContinuousBatcher(eng, eos_ids=(), prefetch=int(os.environ.get("DOTS_BENCH_PREFETCH", str(slots))), admit_group=int(ag) if ag else None)
        reqs_by_page = {}

        def pull_keep(k):
            got = pull(k)
            reqs_by_page.update(got)
            return got
        done = cb.run_pull(pull_keep)
        mine = sorted(done)
        dyn["pages"] = mine
        mixed_meter["decode_steps"] += cb.decode_steps
        mixed_meter["requests"].append([(len(reqs_by_page[p_].input_ids), len(done[p_])) for p_ in mine])
        mixed_meter["last_reqs"], mixed_meter["last_outs"] = [reqs_by_page[p_] for p_ in mine], [done[p_] for p_ in mine]
        out = np.zeros((len(mine), a.max_new_tokens), np.int32)
        out_lens = np.zeros(len(mine), np.int32)
        for i, p_ in enumerate(mine):
            out[i, :len(done[p_])], out_lens[i] = done[p_], len(done[p_])
        t3 = time.perf_counter()
        texts = proc.batch_decode([out[i, :out_lens[i]] for i in range(len(out_lens))])
        host_ms["tokenize_ms"] += t_host[0] * 1e3
        host_ms["preprocess_ms"] += t_host[1] * 1e3
        host_ms["detokenize_ms"] += (time.perf_counter() - t3) * 1e3
        return out, out_lens, texts, [reqs_by_page[p_].input_ids for p_ in mine]

    seq_state = {"k": 0, "sid": 0}

    def step(pipelined=False, sid=None):
        """One batch: tokenise, GPU preprocessing, tower, prefill, greedy decode, detokenise.  pipelined: the tower rows of THIS batch were
        prefetched during the previous step and are taken now; the preprocessing + tower launched here are those of the NEXT batch and
        run on the CU-masked side stream while this batch's decode loop runs on the other CU partition.  Per step the same work either
        way: one preprocessing pass, one tower, one prefill, one decode loop."""
        t0 = time.perf_counter()
        if mixed and a.page_queue:
            return step_page_queue(t0)
        # which page set: an explicit one (the sequential reference batches), else batch k of the rotation; the tower launched by a PIPELINED step is
        # that of batch k + 1, i.e. of the NEXT set
        if sid is None:
            sid = seq_state["k"] % n_sets
            seq_state["k"] += 1
        seq_state["sid"] = sid
        prompts = prompts_of(sid)
        t1 = time.perf_counter()
        if pipelined:
            eng.vit_take()
        grids = preprocess_all((sid + 1) % n_sets if pipelined else sid)
        t2 = time.perf_counter()
        grid = np.asarray(grids, np.int64)
        if pipelined:
            eng.vit_prefetch(pix_dev, grid, on_device=True, after_prefill=True)        # starts behind this batch's prefill
            ids = np.concatenate(prompts)
            lens = np.asarray([len(p) for p in prompts], np.int32)
            out, out_lens = eng.generate(ids, lens, max_new_tokens=a.max_new_tokens, eos_ids=(), vision_taken=True)
        elif mixed:
            from dots_ocr_amd.scheduler import ContinuousBatcher, Request
            reqs, off = [], 0
            for pr, g, n in zip(prompts, grids, n_patches):
                reqs.append(Request(pr, pix[off:off + n], np.asarray([g], np.int64), a.max_new_tokens))
                off += n
            # look-ahead: with EOS disabled and equal caps every slot finishes together, so the whole next group's towers are prefetched
            # round 4: the partition launch plan covers any row count, so the look-ahead now wins at 32 occupied slots (4.04 -> 4.26 pages/s
            # on one GPU, profiles/r04_bench_mixed64*.json) and is the default; DOTS_BENCH_PREFETCH=0 switches it off
            ag = os.environ.get("DOTS_BENCH_ADMIT_GROUP")
            cb = ContinuousBatcher(eng, eos_ids=(), prefetch=int(os.environ.get("DOTS_BENCH_PREFETCH", str(slots))), admit_group=int(ag) if ag else None)
            outs = cb.run(reqs)
            mixed_meter["decode_steps"] += cb.decode_steps
            mixed_meter["requests"].append([(len(r.input_ids), len(o)) for r, o in zip(reqs, outs)])
            mixed_meter["last_reqs"], mixed_meter["last_outs"] = reqs, outs
            out = np.zeros((len(outs), a.max_new_tokens), np.int32)
            out_lens = np.zeros(len(outs), np.int32)
            for i, o in enumerate(outs):
                out[i, :len(o)], out_lens[i] = o, len(o)
        else:
            ids = np.concatenate(prompts)
            lens = np.asarray([len(p) for p in prompts], np.int32)
            out, out_lens = eng.generate(ids, lens, pix_dev, grid, a.max_new_tokens, (), pixel_on_device=True)
        t3 = time.perf_counter()
        texts = proc.batch_decode([out[i, :out_lens[i]] for i in range(len(out_lens))])
        t4 = time.perf_counter()
        host_ms["tokenize_ms"] += (t1 - t0) * 1e3
        host_ms["preprocess_ms"] += (t2 - t1) * 1e3
        host_ms["detokenize_ms"] += (t4 - t3) * 1e3
        return out, out_lens, texts, prompts

    # mixed64: the scheduler drives the engine; meters around the calls it makes give the timed region's roofline inputs without touching the
    # schedule: (a) before a NEW tower is launched the previous one has been taken (done), so its HIP-event sums (flash-attention launches,
    # algorithmic flops) are read then — dots_get_stats only waits for streams that are idle at that point; (b) the wall time from each
    # slots_decode to the poll that follows it (host-synchronised) is the decode time of that chunk, the tower of the look-ahead running beside it
    mixed_meter = {"attn_ms": 0.0, "attn_flops": 0.0, "attn_launches": 0, "vit_ms": 0.0, "vit_flops": 0.0, "towers": 0, "pending": False,
                   "decode_ms": 0.0, "t_dec": None, "decode_steps": 0, "requests": [], "last_reqs": None, "last_outs": None}
    if mixed:
        def harvest():
            if mixed_meter["pending"]:
                st_ = eng.stats()
                mixed_meter["attn_ms"] += st_["vit_attn_ms"]; mixed_meter["attn_flops"] += st_["vit_attn_flops"]
                mixed_meter["attn_launches"] += st_["vit_attn_launches"]; mixed_meter["vit_ms"] += st_["vit_ms"]; mixed_meter["vit_flops"] += st_["vit_flops"]
                mixed_meter["towers"] += 1
                mixed_meter["pending"] = False
        _vf, _vp, _sd, _sp = eng.vit_forward, eng.vit_prefetch, eng.slots_decode, eng.slots_poll

        def vit_forward_m(*aa, **kk):
            harvest(); r_ = _vf(*aa, **kk); mixed_meter["pending"] = True; return r_

        def vit_prefetch_m(*aa, **kk):
            harvest(); r_ = _vp(*aa, **kk); mixed_meter["pending"] = True; return r_

        def slots_decode_m(n_):
            if mixed_meter["t_dec"] is None:
                mixed_meter["t_dec"] = time.perf_counter()
            return _sd(n_)

        def slots_poll_m():
            r_ = _sp()
            if mixed_meter["t_dec"] is not None:
                mixed_meter["decode_ms"] += (time.perf_counter() - mixed_meter["t_dec"]) * 1e3
                mixed_meter["t_dec"] = None
            return r_
        eng.vit_forward, eng.vit_prefetch, eng.slots_decode, eng.slots_poll = vit_forward_m, vit_prefetch_m, slots_decode_m, slots_poll_m
        mixed_meter["harvest"] = harvest

    deep_state = {"k": 0, "queue": [], "last_decode_ms": 0.0, "plen": {}}
    tower_after_prefill = os.environ.get("DOTS_BENCH_TOWER_NOW") != "1"      # =1: the next tower starts beside this batch's prefill (A/B; measured slower)
    trace_steps = os.environ.get("DOTS_BENCH_TRACE") == "1"      # host-side timeline of every pipelined step on stderr
    half_steps = -(-a.max_new_tokens // n_groups)    # decode steps per timed step: a batch gets n_groups x half_steps >= max_new_tokens - 1 of them

    def step_deep():
        """One timed step of the multi-batch pipeline: take the prefetched rows of batch k, preprocess batch k+1 and queue its tower behind
        the prefill, prefill batch k into the free slot group, decode half_steps steps over BOTH groups (batch k: its first half, batch
        k-1: its second half), read batch k-1 out.  Returns batch k-1's tokens (None while the pipeline fills)."""
        k = deep_state["k"]
        t0 = time.perf_counter()
        prompts = prompts_of(k % n_sets)             # batch k = page set k mod n_sets
        t1 = time.perf_counter()
        tr = [("start", t0), ("tokenize", t1)] if trace_steps else None
        if sliced:
            grids = preprocess_all(k % n_sets)
            t2 = time.perf_counter()
            eng.vit_forward(pix_dev, np.asarray(grids, np.int64), on_device=True)       # this batch's tower, whole chip, in stream order before its prefill
        else:
            eng.vit_take()
            if tr: tr.append(("vit_take", time.perf_counter()))
            grids = preprocess_all((k + 1) % n_sets)     # the tower queued now is batch k + 1's
            t2 = time.perf_counter()
            eng.vit_prefetch(pix_dev, np.asarray(grids, np.int64), on_device=True, after_prefill=tower_after_prefill)
        if tr: tr.append(("preprocess+prefetch", time.perf_counter()))
        group = [(k % n_groups) * B + i for i in range(B)]
        eng.slots_prefill(group, np.concatenate(prompts), [len(p) for p in prompts], [a.max_new_tokens] * B)
        for s_, p_ in zip(group, prompts):
            deep_state["plen"][s_] = len(p_)
        if tr: tr.append(("slots_prefill", time.perf_counter()))
        eng.slots_poll()                             # the prefill is done (synchronises with the main stream's chain only, not with the tower):
        td = time.perf_counter()                     # the decode chunks are timed from here
        done_steps = 0
        while done_steps < half_steps:               # chunks: the decode stream (partition beside the tower / whole chip after it) is picked per chunk
            n = min(64, half_steps - done_steps)
            eng.slots_decode(n)
            done_steps += n
        if tr: tr.append(("decode issued", time.perf_counter()))
        fin, lens_ = eng.slots_poll()                # synchronises with the decode chain only (the tower of the next batch may still run)
        deep_state["last_decode_ms"] = (time.perf_counter() - td) * 1e3
        if tr: tr.append(("poll", time.perf_counter()))
        out = out_lens = texts = None
        deep_state["queue"].append(group)
        prev = deep_state["queue"].pop(0) if len(deep_state["queue"]) == n_groups else None
        t3 = time.perf_counter()
        if prev is not None:
            assert all(fin[s_] == 1 for s_ in prev), "a batch did not finish within its two half loops"
            out = np.zeros((B, a.max_new_tokens), np.int32)
            out_lens = np.zeros(B, np.int32)
            for i, s_ in enumerate(prev):
                toks = eng.slot_read(s_, a.max_new_tokens)
                out[i, :len(toks)], out_lens[i] = toks, len(toks)
                eng.slot_release(s_)
            texts = proc.batch_decode([out[i, :out_lens[i]] for i in range(B)])
        t4 = time.perf_counter()
        if tr:
            tr.append(("read-out", t4))
            print("[step %d] " % k + "  ".join("%s +%.1f" % (nm, (tt - t0) * 1e3) for nm, tt in tr), file=sys.stderr, flush=True)
        deep_state["k"] = k + 1
        seq_state["sid"] = (k - n_groups + 1) % n_sets           # the batch read out now
        host_ms["tokenize_ms"] += (t1 - t0) * 1e3
        host_ms["preprocess_ms"] += (t2 - t1) * 1e3
        host_ms["detokenize_ms"] += (t4 - t3) * 1e3
        return out, out_lens, texts, prompts

    seq_stats = None
    seq_out = None                                   # tokens of the strictly sequential batch: the pipelined steps decode the SAME pages
    pipelined_outs = []                              # (out, out_lens) of every pipelined step, compared after the timed region
    if overlap:
        # one strictly sequential batch first: warms everything up AND gives the per-kernel whole-chip timings of this very run
        # (reported beside the timed region's, where the tower and the decode loop share the chip); then the pipeline is primed
        o0, l0, _, seq_prompts = step(sid=0)
        seq_out = [(o0.copy(), l0.copy())]
        seq_stats = eng.stats()
        for s_ in range(1, n_sets):                  # the other page sets' sequential reference tokens (untimed)
            o_, l_, _, _ = step(sid=s_)
            seq_out.append((o_.copy(), l_.copy()))
        if deep:
            eng.set_eos([])
            eng.slots_reset()                        # sequence-slot mode: every slot free, every KV page in the pool (drops any pending prefetch)
        if deep and not sliced and "DOTS_OCR_TOWER_TAIL_LAYERS" not in os.environ:
            eng.tower_tail(-1)                       # a step's decode work is finite here (half_steps): the tower's last blocks take the whole chip once it has drained
        if not sliced:
            eng.vit_prefetch(pix_dev, np.asarray(preprocess_all(0), np.int64), on_device=True)       # batch 0 = set 0
        if deep:
            for _ in range(n_groups - 1):
                step_deep()                          # fills the pipeline (no batch completes yet); untimed
    run_step = step_deep if deep else (lambda: step(overlap))
    for _ in range(a.warmup):
        o_, l_, _, _ = run_step()
        if overlap and o_ is not None:
            pipelined_outs.append((o_.copy(), l_.copy(), seq_state["sid"]))
    for k in host_ms:
        host_ms[k] = 0.0
    if mixed:
        mixed_meter["harvest"]()
        for k_ in ("attn_ms", "attn_flops", "vit_ms", "vit_flops", "decode_ms"):
            mixed_meter[k_] = 0.0
        mixed_meter.update({"attn_launches": 0, "towers": 0, "decode_steps": 0, "requests": []})
    eng.synchronize(); torch.cuda.synchronize(); barrier()
    t0 = time.perf_counter()
    phase = {"vit_ms": 0.0, "prefill_ms": 0.0, "decode_ms": 0.0, "vit_attn_ms": 0.0}
    last = None
    deep_bytes_per_step = 0.0
    if deep:
        # a static batch's decode_bytes = steps x W + kv_tok x sum over rows and steps of (context + 1): solve for W (every weight byte once per step)
        kv_tok = cfg.num_hidden_layers * cfg.num_key_value_heads * 128 * 2 * 2
        L0, st_n = len(seq_prompts[0]), seq_stats["decode_steps"]
        W = (seq_stats["decode_bytes"] - kv_tok * sum(len(p) * st_n + st_n * (st_n + 1) // 2 for p in seq_prompts)) / st_n
        ctx_sum = 0
        for p in seq_prompts:                        # per timed step every page contributes its first half (steps 1..h) in one group and its second half in the other
            ctx_sum += sum(len(p) + t + 1 for t in range(min(n_groups * half_steps, a.max_new_tokens - 1)))
        deep_bytes_per_step = half_steps * W + kv_tok * ctx_sum
    for _ in range(a.steps):
        out, out_lens, texts, prompts = run_step()
        if overlap:
            pipelined_outs.append((out, out_lens, seq_state["sid"]))   # a reference (fresh arrays every step): compared after the timed region
        st = eng.stats()                             # device-side HIP-event times of this step (static batches only)
        if deep:                                     # slot mode records no decode events: wall time of this step's decode chunks (host-synchronised),
            st = dict(st)                            # algorithmic bytes by the formula the engine uses for a static batch (weights once per step + KV read)
            st["decode_ms"] = deep_state["last_decode_ms"]
            st["decode_steps"] = half_steps
            st["decode_bytes"] = deep_bytes_per_step
            st["prefill_flops"] = seq_stats["prefill_flops"]
        for k in phase:
            phase[k] += st[k]
        last = st
    eng.synchronize(); torch.cuda.synchronize()
    if mixed:
        mixed_meter["harvest"]()
    # the only data-path collective of the job — the gather of the generated token ids on rank 0 (dp.py: two all_gathers over RCCL) —
    # is INSIDE the timed region, timed on its own as well
    tg = time.perf_counter()
    if mixed and a.page_queue:                       # the pages this rank drew in the last job (any subset of the job; the gather restores page order)
        my_pages = dyn["pages"]
        sizes = [sizes_all[p_] for p_ in my_pages]
    if not mixed:
        my_pages = sets[seq_state["sid"]]            # the set the last step's tokens belong to
    gathered = dp.gather_token_ids(out, out_lens, page_index=my_pages)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - tg) * 1e3
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    # ---- the decode step the timed region replays, ALONE on the chip (outside the timed region, rank 0): one more batch is admitted with no
    # look-ahead tower behind it, so every slot group holds rows; 64 graph replays on the whole chip, host-synchronised on both sides
    dec_alone_rows = None
    if deep and not sliced and not mixed and rank == 0 and os.environ.get("DOTS_BENCH_DECODE_ALONE", "1") != "0":
        try:                                         # an optional extra: whatever goes wrong here must not cost the run its line
            k_ = deep_state["k"]
            eng.vit_take()
            prompts_ = prompts_of(k_ % n_sets)
            group_ = [(k_ % n_groups) * B + i for i in range(B)]
            eng.slots_prefill(group_, np.concatenate(prompts_), [len(p_) for p_ in prompts_], [a.max_new_tokens] * B)
            for s_, p_ in zip(group_, prompts_):
                deep_state["plen"][s_] = len(p_)
            fin_, lens_ = eng.slots_poll()
            eng.synchronize(); torch.cuda.synchronize()      # nothing else is queued: no tower, no prefill
            live_ = [s_ for s_ in range(len(fin_)) if fin_[s_] == 0]
            n_alone = min([64] + [a.max_new_tokens - 1 - int(lens_[s_]) for s_ in live_])      # no row may finish inside the measurement
            ta = time.perf_counter()
            if n_alone >= 8:
                eng.slots_decode(n_alone)
                eng.slots_poll()
            ms_alone = (time.perf_counter() - ta) * 1e3 / max(1, n_alone)
            kv_tok_ = cfg.num_hidden_layers * cfg.num_key_value_heads * 128 * 2 * 2
            ctx0 = [deep_state["plen"][s_] + int(lens_[s_]) for s_ in live_]       # tokens in the KV cache of each row before the first of the n_alone steps
            bytes_alone = W + kv_tok_ * sum(c_ + (n_alone - 1) / 2.0 + 1 for c_ in ctx0)     # mean over the steps: every weight byte once + the KV read
            dec_alone_rows = {"bound": "hbm", "achieved": bytes_alone / (ms_alone * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "rows": len(live_),
                              "ms_per_step": ms_alone, "steps": n_alone, "mean_context": float(np.mean(ctx0)) + n_alone / 2.0, "algorithmic_bytes_per_step": bytes_alone,
                              "cus": 256, "note": "the decode step of the timed region (every slot group holding rows) replayed alone on the whole chip after the timed region: "
                                                  "no tower, no prefill; wall time of %d hipGraph replays, host-synchronised on both sides" % n_alone}
            dec_alone_rows["frac"] = dec_alone_rows["achieved"] / dec_alone_rows["peak"]
            if n_alone < 8:                              # short generations (--max-new-tokens below the pipeline's depth): nothing to replay
                dec_alone_rows = None
        except Exception as exc:
            dec_alone_rows = None
            print(f"bench.py: decode-alone measurement skipped: {exc!r}", file=sys.stderr, flush=True)
    # ---- parity of what was timed: every pipelined step (CU-masked side stream, half-chip decode plan, deferred tower)
    # decoded the same pages with the same prompts as the strictly sequential batch this run started with — the tokens must be
    # identical bit for bit, on every rank, or the run fails.
    parity = None
    if overlap and seq_out is not None:
        for k_step, (o_, l_, sid_) in enumerate(pipelined_outs):
            if not (np.array_equal(l_, seq_out[sid_][1]) and np.array_equal(o_, seq_out[sid_][0])):
                bad = np.argwhere(o_ != seq_out[sid_][0])
                raise SystemExit(f"bench.py: rank {rank}: pipelined step {k_step} (page set {sid_}) produced different tokens than the sequential batch of the same pages "
                                 f"(first difference at page {int(bad[0][0])}, token {int(bad[0][1])}): the timed configuration is NOT parity-clean")
        parity = {"parity_vs_sequential": "bitwise", "steps_checked": len(pipelined_outs),
                  "tokens_per_step_checked": int(seq_out[0][1].sum()), "page_sets": n_sets,
                  "page_sets_note": f"step k processes page set k mod {n_sets} (distinct pages, text and prompts; all sets resident in HBM); every step's tokens are compared "
                                    "with the strictly sequential batch of ITS set" if n_sets > 1 else "every step processes the same pages"}
        if n_sets > 1:
            assert len({sid_ for _, _, sid_ in pipelined_outs}) == min(n_sets, len(pipelined_outs)), "the steps did not rotate through the page sets"
            assert not np.array_equal(prompts_of(0)[0], prompts_of(1)[0]) and not np.array_equal(set_arrays[0][0], set_arrays[1][0]), "two page sets hold the same inputs"
            # (their TOKENS may coincide: seeded random weights fall into one repeated token within a few steps whatever the page shows)
    n_ranks = 1
    per_rank = None
    if use_dist:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)                        # how many ranks RCCL really sees
        n_ranks = int(ones.item())
        # diagnosis of a multi-GPU run (outside the timed region, one KB-sized gather): every rank's own step time, host phases, device
        # phases and set-up time, so that a sub-linear result names its cause (a slow rank, host contention, set-up skew)
        names = ["ms_per_step", "tokenize_ms", "preprocess_ms", "detokenize_ms", "vit_ms", "prefill_ms", "decode_ms", "setup_s"]
        mine = torch.tensor([dt_local / a.steps * 1e3, host_ms["tokenize_ms"] / a.steps, host_ms["preprocess_ms"] / a.steps, host_ms["detokenize_ms"] / a.steps,
                             phase["vit_ms"] / a.steps, phase["prefill_ms"] / a.steps, phase["decode_ms"] / a.steps, setup_s], device="cuda", dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        tab = torch.stack(allv).cpu().numpy()        # [rank, quantity]
        per_rank = {n: {"min": float(tab[:, i].min()), "max": float(tab[:, i].max()), "rank_of_max": int(tab[:, i].argmax()),
                        "all": [round(float(x), 3) for x in tab[:, i]]} for i, n in enumerate(names)}

    if rank == 0:
        K = a.steps
        pages_total = n_job_pages * K
        new_tok = sum(len(t) for _, t in gathered) * K
        res = {
            "metric": (f"pages/sec, dots.ocr 1.7B {'fp8 weights' if fp8 else 'bf16'}, chart->SVG page (preprocess + ViT + prefill + {a.max_new_tokens}-token greedy decode + detokenise)"
                       if a.workload == "svg" else
                       f"pages/sec, dots.ocr 1.7B {'fp8' if fp8 else 'bf16'}, "
                       f"{ {'a4': 'A4@200dpi page batch', 'mixed64': '64 mixed-size pages (continuous batching)', 'highres': 'high-res 1344x1344 page batch', 'tiny': 'tiny-dims plumbing run'}[a.workload] } "
                       f"(preprocess + ViT + prefill + {a.max_new_tokens}-token greedy decode + detokenise)"),
            "value": pages_total / dt, "unit": "pages/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong" if mixed else "weak", "vs_baseline": None,
            "dtype": "fp8 e4m3 weights (per-output-channel fp32 scale) x bf16 activations, fp32 accumulate" if fp8 else "bf16", "data": "synthetic pages (PIL text lines) as uint8 pixels in HBM, seeded random weights at the checkpoint's dimensions",
            "output_tok_s": new_tok / dt, "rccl_ranks": n_ranks, "gathered_pages": len(gathered), "gather_ms": gather_ms,
            "gather_note": "token-id gather on rank 0 (the job's only collective; 2 all_gathers), once per job, inside the timed region", "setup_s": setup_s,
        }
        if parity:
            res.update(parity)
        if per_rank:
            res["per_rank"] = per_rank
        if mixed:
            from collections import Counter
            mm = mixed_meter
            kv_tok = cfg.num_hidden_layers * cfg.num_key_value_heads * 128 * 2 * 2
            Wb = 2.0 * (sum(int(np.prod(v_.shape)) for k_, v_ in sd.items() if k_.startswith("model.layers.") and v_.dim() == 2) + cfg.vocab_size * cfg.hidden_size)   # 3 087 138 816 B
            if fp8:
                Wb /= 2.0
            # algorithmic decode bytes of the timed region: every weight byte once per decode step + the KV read of every row and step
            kv_bytes = sum(kv_tok * sum(L_ + t_ for t_ in range(1, n_)) for job in mm["requests"] for (L_, n_) in job)
            dec_bytes = mm["decode_steps"] * Wb + kv_bytes
            attn_tf = mm["attn_flops"] / (mm["attn_ms"] / 1e3) / 1e12 if mm["attn_ms"] > 0 else 0.0
            dec_gbs = dec_bytes / (mm["decode_ms"] / 1e3) / 1e9 if mm["decode_ms"] > 0 else 0.0
            dec_cus_m = int(os.environ.get("DOTS_OCR_OVERLAP_DEC_CUS", "64")) // 8 * 8
            res["roofline"] = {"bound": "mfma", "kernel": "flash_attn64_kernel (ViT bidirectional var-len attention) over the timed region's towers: ragged packed batches of the six page sizes",
                               "achieved": attn_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": attn_tf / PEAK_BF16_TFLOPS, "traffic": None,
                               "algorithmic_flops": mm["attn_flops"], "launches": mm["attn_launches"], "towers": mm["towers"],
                               "sum_of_launch_ms": mm["attn_ms"],
                               "note": f"HIP-event sums on the stream each tower ran on; look-ahead towers run on the {256 - dec_cus_m}-CU partition beside the decode loop "
                                       f"(frac is against the WHOLE chip's peak: x 256 / {256 - dec_cus_m} for the partition's), the first group's tower on the whole chip"}
            res["roofline_decode"] = {"bound": "hbm", "achieved": dec_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": dec_gbs / PEAK_HBM_GBS, "traffic": None,
                                      "decode_steps": mm["decode_steps"], "algorithmic_bytes": dec_bytes, "decode_wall_ms": mm["decode_ms"],
                                      "note": f"wall time from each slots_decode chunk to the poll after it (host-synchronised), up to {slots} rows per step, on the {dec_cus_m}-CU "
                                              "partition while a look-ahead tower runs and on the whole chip otherwise"}
            res["phase_ms_per_step"] = {"vit_ms": mm["vit_ms"] / K, "vit_attn_ms": mm["attn_ms"] / K, "decode_ms": mm["decode_ms"] / K, **{k: val / K for k, val in host_ms.items()}}
            # parity of what was timed: 8 of this rank's pages (every size class first, then the front of the list) decoded again ALONE through the
            # static single-sequence generate — the continuous batch with look-ahead towers must have produced the same tokens, bit for bit
            seen, sample = set(), []
            for i_, sz_ in enumerate(sizes):
                if sz_ not in seen:
                    seen.add(sz_); sample.append(i_)
            sample = (sample + [i_ for i_ in range(len(sizes)) if i_ not in sample])[:8]
            eng.vit_forward, eng.vit_prefetch, eng.slots_decode, eng.slots_poll = _vf, _vp, _sd, _sp
            bad = []
            for i_ in sample:
                r_ = mm["last_reqs"][i_]
                single, sl = eng.generate(r_.input_ids, np.asarray([len(r_.input_ids)], np.int32), r_.pixel_values.data_ptr(), np.asarray(r_.grid_thw, np.int64),
                                          a.max_new_tokens, (), pixel_on_device=True)
                if not np.array_equal(single[0, :sl[0]], mm["last_outs"][i_]):
                    bad.append(my_pages[i_])
            if bad:
                raise SystemExit(f"bench.py: mixed64: pages {bad} decoded differently in the continuous batch than alone: the timed configuration is NOT parity-clean")
            res["parity_vs_single_sequence"] = "bitwise"
            res["pages_checked"] = [int(my_pages[i_]) for i_ in sample]
            res["tokens_per_page_checked"] = a.max_new_tokens
            res["config"] = {"workload": "mixed64: 64 pages " + ", ".join(f"{n}x {w}x{h}" for (w, h), n in sorted(Counter(sizes_all).items())) +
                                         f"; cost-sharded (LPT) over {world} rank(s), continuous batching over {slots} slots per rank, "
                                         f"max_new_tokens={a.max_new_tokens}, EOS disabled",
                             "pages_per_gpu": [len(s) for s in shards], "slots_per_gpu": slots,
                             "occupied_slots_per_gpu_at_start": [min(slots, len(s)) for s in shards], "parallelism": f"dp{world}"}
            if a.page_queue:
                res["config"]["page_queue"] = ("--page-queue: NOT the static shard described above — the ranks pulled page indices from one shared queue (dp.PageQueue, "
                                               "costliest first, a fetch-and-add on the process group's host-side store) as their slots drained; "
                                               f"rank 0 drew {len(my_pages)} pages in the last job")
            occ = max(min(slots, len(s)) for s in shards)
            res["scaling_note"] = (
                f"strong scaling of a FIXED 64-page job: a rank holding {max(len(s) for s in shards)} pages decodes at most {occ} sequences at a time, and decode "
                "is latency-bound (a step costs nearly the same for 8 as for 32 rows), so the per-GPU rate falls with the slot occupancy, not with "
                "the interconnect (the only collective is the final KB-sized gather).  1-GPU rates of this job by slot count, measured "
                "(profiles/r02_bench_mixed64_*.json, bench.py --workload mixed64 --batch S): 2.90 / 3.40 / 3.92 / 4.06 pages/s at S = 8 / 16 / 32 / 64. "
                "The like-for-like 1-GPU denominator for an 8-rank run (8 pages per rank) is the S = 8 figure, not the S = 32 default.")
        else:
            res["config"] = {"workload": f"{a.workload}: {B} pages/GPU of {size[0]}x{size[1]} px -> {n_patches[0]} patches, "
                                         f"{len(prompts[0])} prompt tokens/page, max_new_tokens={a.max_new_tokens}, EOS disabled",
                             "pages_per_gpu": B, "parallelism": f"dp{world}", "decode_rows_in_flight": rif if deep else B}
            def recorded(name, key):            # PMC numbers come from separate rocprofv3 --pmc passes over this same command
                f = ROOT / "profiles" / name
                if a.workload == "a4" and B == 8 and f.exists():
                    return json.loads(f.read_text()).get(key)
                return None

            def rooflines(ph, n, lastst, cus_vit, cus_dec):
                """roofline objects from per-phase HIP-event sums `ph` over `n` steps.  `frac` is ALWAYS against the whole chip's peak; when the
                kernel ran on a CU partition (overlapped mode) its share of the chip is stated and the fraction of that share's peak added."""
                attn_s, dec_s, vit_s = ph["vit_attn_ms"] / 1e3, ph["decode_ms"] / 1e3, ph["vit_ms"] / 1e3
                attn_tflops = lastst["vit_attn_flops"] * n / attn_s / 1e12 if attn_s > 0 else 0.0
                dec_gbs = lastst["decode_bytes"] * n / dec_s / 1e9 if dec_s > 0 else 0.0
                vit_tflops = lastst["vit_flops"] * n / vit_s / 1e12 if vit_s > 0 else 0.0
                r = {"bound": "mfma", "kernel": "flash_attn64_kernel (ViT bidirectional var-len attention; 4 waves x 64 query rows, round 4)",
                     "achieved": attn_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": attn_tflops / PEAK_BF16_TFLOPS,
                     "traffic": recorded("r06_flash_attn_traffic.json", "traffic_bytes_per_launch"),
                     "traffic_unit": "bytes/launch (PMC on the whole chip, profiles/r06_flash_attn_traffic.json: 4.31 GB against 1.95 GB compulsory, unchanged since round 4)",
                     "algorithmic_flops_per_launch": lastst["vit_attn_flops"] / max(1, lastst["vit_attn_launches"]),
                     "launches_per_step": lastst["vit_attn_launches"],
                     "avg_launch_ms": ph["vit_attn_ms"] / n / max(1, lastst["vit_attn_launches"])}
                rv = {"bound": "mfma", "achieved": vit_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": vit_tflops / PEAK_BF16_TFLOPS}
                rd = {"bound": "hbm", "achieved": dec_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": dec_gbs / PEAK_HBM_GBS,
                      "ms_per_decode_step": ph["decode_ms"] / n / max(1, lastst["decode_steps"]),
                      "algorithmic_bytes_per_decode_step": lastst["decode_bytes"] / max(1, lastst["decode_steps"]),
                      "traffic": recorded("r04_decode_traffic.json", "traffic_bytes_per_decode_step"),
                      "traffic_unit": "bytes per decode step (PMC on the whole chip, B = 8, profiles/r04_decode_traffic.json)"}
                if cus_vit < 256:
                    for o in (r, rv):
                        o["cus"] = cus_vit
                        o["frac_of_partition_peak"] = o["frac"] * 256.0 / cus_vit
                        o["note"] = (f"timed region = software-pipelined batches: this kernel runs on a {cus_vit}-CU partition while the other {256 - cus_vit} CUs replay the "
                                     "decode loop of the previous batch, so `frac` (against the WHOLE chip's peak) is by construction about half of what the same "
                                     "kernel reaches alone on the chip — that figure, measured in the sequential batch this run starts with, is in roofline_sequential / "
                                     "roofline_vit_sequential; whole-step utilisation is in roofline_step")
                    rd["note"] = ("timed region = software-pipelined batches: most decode steps run on the decode partition beside the next batch's vision tower "
                                  "(half-chip launch plan); the step alone on the whole chip is in roofline_decode_sequential")
                    rd["cus"] = f"{cus_dec} while the next batch's tower runs, 256 after it"
                    if deep:                       # the B = 8 whole-chip figure describes roofline_decode_sequential; the 64-row partition step has its own PMC pass (round 5)
                        rd["traffic"] = recorded("r06_decode_traffic_64rows.json", "traffic_bytes_per_decode_step") if rif == 64 and a.workload == "a4" else None
                        rd["traffic_unit"] = ("bytes per decode step (PMC, 64 rows on the 64-CU partition plan, profiles/r06_decode_traffic_64rows.json: 1.09 x the "
                                              "algorithmic bytes (round 5: 1.24) — gate|up and lm_head now pass their weights through the CUs once; what is left are "
                                              "the X images the projections re-read, the K-quarter partials of down_proj and the attention partials)"
                                              if rd["traffic"] else "not measured for the %d-row step (PMC traffic at B = 8: roofline_decode_sequential.traffic)" % rif)
                return r, rv, rd
            dec_cus = int(os.environ.get("DOTS_OCR_OVERLAP_DEC_CUS", "128")) // 8 * 8
            cv, cd = (256 - dec_cus, dec_cus) if overlap and not sliced else (256, 256)
            res["roofline"], res["roofline_vit"], res["roofline_decode"] = rooflines(phase, K, last, cv, cd)
            res["decode_tok_s"] = int(out_lens.sum()) * K / (phase["decode_ms"] / 1e3) if phase["decode_ms"] > 0 else None
            res["phase_ms_per_step"] = {**{k: val / K for k, val in phase.items()}, **{k: val / K for k, val in host_ms.items()}}
            if overlap:
                res["overlap"] = {
                    "mode": "software-pipelined batches: the vision tower of batch k+1 (stream masked to CUs %d-255, an equal share of every XCD) runs "
                            "while batch k is prefilled and decoded (decode graph on the stream masked to CUs 0-%d until the tower is done, whole chip "
                            "after); per step exactly one preprocessing pass, one tower, one prefill, one decode loop; phase_ms_per_step therefore "
                            "OVERLAP and sum to more than ms_per_step" % (dec_cus, dec_cus - 1),
                    "dec_cus": 256 if sliced else dec_cus, "vit_cus": 256 if sliced else 256 - dec_cus,
                    "tower_tail_blocks": None if sliced else eng.tower_tail(),
                    "tower_tail_note": "the last tower_tail_blocks of the 42 tower blocks (and the merger) run on the whole chip, not on the tower partition: the "
                                       "engine sizes the partition part from the previous step's events so that it ends with the decode loop (dots_tower_tail; "
                                       "partitions cannot be re-balanced finer than 32 CUs: with 56 decode CUs every decode kernel runs at half speed)",
                    "time_sliced": ("no CU partitions: the tower of the new batch, its prefill and the decode steps over all rows in flight take turns on the "
                                    "whole chip (the 'mode' text above describes the partitioned default)") if sliced else None,
                    "batches_decoding_together": rif // B if deep else 1,
                    "multi_batch_pipeline": ("continuous batching over %d sequence slots with admission in groups of %d pages: a timed step takes the prefetched tower rows "
                                           "of batch k, queues batch k+1's tower behind the prefill, prefills batch k into the free slot group and runs %d decode steps "
                                           "over ALL groups (batch k: tokens 1-%d, the older batches their later parts), then reads the oldest batch out — every weight "
                                           "byte is streamed once per %d rows instead of once per %d; a page's tokens do not depend on what shares its decode step "
                                           "(bitwise check below)" % (rif, B, half_steps, half_steps, rif, B)) if deep else None,
                    "sequential_step_ms_same_run": seq_stats["total_ms"] if seq_stats else None,
                    "why": "two unmasked streams time-slice the chip (measured: no overlap); with complementary CU masks the MFMA-bound tower and the "
                           "latency-bound decode loop run side by side (tools/overlap_probe.py, profiles/r03_overlap_probe.txt)"}
                # the same kernels alone on the whole chip: the strictly sequential batch this run started with (1 step: 42 attention launches, 1023 decode steps)
                seq_phase = {k: seq_stats[k] for k in phase}
                rs, rvs, rds = rooflines(seq_phase, 1, seq_stats, 256, 256)
                res["roofline_sequential"], res["roofline_vit_sequential"], res["roofline_decode_sequential"] = rs, rvs, rds
                step_s = dt / K
                res["roofline_step"] = {"note": "sustained over the WHOLE step (all phases, overlapped or not): algorithmic ViT+prefill flops / step time vs the "
                                                "dense bf16 peak, and algorithmic decode bytes / step time vs the HBM peak",
                                        "mfma_frac": (last["vit_flops"] + last["prefill_flops"]) / step_s / 1e12 / PEAK_BF16_TFLOPS,
                                        "hbm_frac": last["decode_bytes"] / step_s / 1e9 / PEAK_HBM_GBS}
            if overlap and seq_stats:
                # the three shapes of the same work side by side: throughput AND how long a page waits for its tokens
                step_ms = dt / K * 1e3
                one = None
                f1 = ROOT / "profiles" / "r06_bench_a4_one_batch_in_flight.json"
                if a.workload == "a4" and B == 8 and f1.exists():
                    j1 = json.loads(f1.read_text())
                    one = {"pages_per_s": j1["value"], "page_latency_s": 2 * j1["ms_per_step"] / 1e3,
                           "source": "profiles/r06_bench_a4_one_batch_in_flight.json (bench.py --rows-in-flight 8 on the same build: the tower of batch k+1 beside the decode loop of batch k)"}
                res["throughput_shapes"] = {
                    "headline": f"pipelined, {rif if deep else B} rows in flight",
                    f"pipelined_{rif if deep else B}_rows_in_flight": {"pages_per_s": pages_total / dt, "page_latency_s": (n_groups if deep else 2) * step_ms / 1e3,
                                                                         "note": "a page's tokens are complete this many seconds after its batch was admitted (throughput configuration: "
                                                                                 "the reference's parse_pdf over a document)"},
                    "one_batch_in_flight": one,
                    "sequential_batch": {"pages_per_s": n_job_pages / world / (seq_stats["total_ms"] / 1e3) * world, "page_latency_s": seq_stats["total_ms"] / 1e3,
                                         "source": "the strictly sequential batch this run starts with (tower -> prefill -> 1023 decode steps on the whole chip, nothing overlapped)"}}
            # north_star targets, stated as what the evidence supports: the kernels ALONE on the chip (the sequential batch of this run) and
            # what the chip sustains over the whole timed step
            vit_alone = res.get("roofline_vit_sequential", res["roofline_vit"])["frac"]
            dec_alone = res.get("roofline_decode_sequential", res["roofline_decode"])["frac"]
            step_s = dt / K
            res["targets"] = {"vit_mfma_frac_alone": vit_alone, "decode_hbm_frac_alone": dec_alone,
                              "step_mfma_frac": (last["vit_flops"] + last["prefill_flops"]) / step_s / 1e12 / PEAK_BF16_TFLOPS,
                              "step_hbm_frac": last["decode_bytes"] / step_s / 1e9 / PEAK_HBM_GBS,
                              "north_star": {"vit_mfma_frac": 0.40, "decode_hbm_frac": 0.50},
                              "met": {"vit_mfma": bool(vit_alone >= 0.40), "decode_hbm": bool(dec_alone >= 0.50)}}
            if dec_alone_rows is not None:          # the timed region's own decode step (all slot groups full) alone on the chip, measured after the timed region
                res["roofline_decode_alone_rows_in_flight"] = dec_alone_rows
                res["targets"]["decode_hbm_frac_alone_rows_in_flight"] = dec_alone_rows["frac"]
                res["targets"]["decode_hbm_basis"] = ("decode_hbm_frac_alone / met.decode_hbm: B = %d (the sequential batch); decode_hbm_frac_alone_rows_in_flight: the "
                                                      "%d-row step the timed region replays, alone on the chip" % (B, dec_alone_rows["rows"]))
                res["targets"]["met"]["decode_hbm_rows_in_flight"] = bool(dec_alone_rows["frac"] >= 0.50)
            if a.workload == "svg":                 # 4096 decode steps at B = 1 dominate this configuration: its roofline is the HBM one
                res["roofline_vit_attn"] = res["roofline"]
                res["roofline"] = {**res["roofline_decode"], "kernel": "one decode step (dec_qkv / decode_attn / combine / dec_proj / dec_gateup x 28 + dec_lmhead)"}
        if not mixed:
            # the reference's inputs.to("cuda") (dots_ocr/parser.py:107): `value` is measured with the uint8 pages resident in HBM (the contract of this
            # bench); the upload of one step's pages is timed here, outside the timed region, and the PCIe-inclusive rate stated beside it
            h2d = []
            for _ in range(3):
                th = time.perf_counter()
                for dptr, arr in zip(set_dev[0], set_arrays[0]):
                    eng.copy_to_device(dptr, arr)
                h2d.append((time.perf_counter() - th) * 1e3)
            nbytes = sum(arr.nbytes for arr in set_arrays[0])
            res["h2d"] = {"ms_per_step": min(h2d), "bytes_per_step": nbytes, "gb_per_s": nbytes / (min(h2d) / 1e3) / 1e9,
                          "pcie_inclusive_pages_per_s": n_job_pages / (dt / K + min(h2d) / 1e3) if world == 1 else None,
                          "note": "uint8 pixels of one step's pages, pageable host memory -> HBM, blocking copies, best of 3, NOT inside the timed region and not hidden "
                                  "behind compute in the PCIe-inclusive figure (worst case: the pipeline could upload batch k+1 while batch k runs)"}
        eng.close()
        if world == 1 and a.workload == "a4" and not a.no_other_configs and B == 8 and a.max_new_tokens == 1024 and not a.no_overlap and os.environ.get("DOTS_BENCH_OTHER", "1") != "0":
            res["other_configs"] = other_config_legs()
        if world == 1 and not a.no_cpu_baseline and a.workload == "a4":
            cores = min(os.cpu_count() or 1, 64)
            res["cpu_baseline"] = cpu_baseline(cfg, sd, cores, pages[0], prompts[0], a.max_new_tokens)
        print(json.dumps(res), flush=True)
    eng.close()
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
