#!/usr/bin/env python3
"""Headline benchmark: pages/sec + output tok/s, dots.ocr 1.7B bf16, A4@200dpi page batch
(BASELINE.json metric; workload = configs[1]: batch of 8 synthetic A4 pages per GPU, layout-all
prompt shape, greedy, max_new_tokens=1024 with EOS disabled so every page emits exactly 1024 tokens).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over one batch: ViT encode + merger + LM prefill + the
decode loop, inputs (normalised patches, token ids) already resident in HBM.  Data-parallel: one
process per GPU, a full replica each, the page batch sharded by page, the only collective being
the final gather of token ids (RCCL).  scaling = weak (8 pages per GPU).

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
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak (same guide; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="pages per GPU per step")
    ap.add_argument("--max-new-tokens", type=int, default=1024)
    ap.add_argument("--workload", default="a4", choices=["a4", "highres", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(cfg, sd_bf16, cores):
    """The CPU oracle (fp32 PyTorch restatement of the reference's HF path) timed on a bounded sample of
    the same workload: ONE synthetic page at quarter linear scale, 16 greedy tokens.  Checker code used
    only as the baseline being reported, never on the product path."""
    from dots_ocr_amd.image_utils import preprocess_image
    from dots_ocr_amd.synthetic import A4_200DPI, synth_page, synth_prompt_ids
    from oracle import model as om
    torch.set_num_threads(cores)
    sd = {k: v.float() for k, v in sd_bf16.items()}
    page = synth_page(0, (A4_200DPI[0] // 4, A4_200DPI[1] // 4))
    pv, thw = preprocess_image(page)
    ids = torch.from_numpy(synth_prompt_ids(cfg, thw[1] * thw[2] // 4).astype(np.int64))
    n_new = 16
    t0 = time.perf_counter()
    om.generate(sd, cfg, ids, torch.from_numpy(pv), torch.tensor([thw]), n_new, emulate_bf16=False)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "pages/s", "cores": cores, "kind": "port",
            "sample": f"1 synthetic page {page.size[0]}x{page.size[1]} px ({pv.shape[0]} patches, {len(ids)} prompt tokens), "
                      f"{n_new} greedy tokens, full-size 1.2B ViT + 1.7B LM in fp32: {dt:.1f} s, {n_new / dt:.2f} tok/s incl. prefill"}


def main():
    a = parse()
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
    from dots_ocr_amd.image_utils import preprocess_image
    from dots_ocr_amd.synthetic import A4_200DPI, HIGH_RES, synth_page, synth_prompt_ids
    from dots_ocr_amd.weights import random_state_dict

    if a.workload == "tiny":
        cfg, size = DotsConfig.tiny(layers=4, v_layers=4), (420, 588)
    else:
        cfg, size = DotsConfig(), (A4_200DPI if a.workload == "a4" else HIGH_RES)
    B = a.batch
    t_setup = time.perf_counter()
    sd = random_state_dict(cfg, seed=a.seed, threads=min(32, os.cpu_count() or 8))

    # ---- this rank's shard of the page batch (weak scaling: B pages per GPU), preprocessed on the host
    pages = [synth_page(rank * B + i, size) for i in range(B)]
    feats, grids = zip(*(preprocess_image(p) for p in pages))
    pv = np.concatenate(feats, 0)
    grid = np.asarray(grids, np.int64)
    n_vis = [int(g[1] * g[2] // 4) for g in grid]
    prompts = [synth_prompt_ids(cfg, n, seed=rank * B + i) for i, n in enumerate(n_vis)]
    ids = np.concatenate(prompts)
    lens = np.asarray([len(p) for p in prompts], np.int32)
    max_seq = int(lens.max()) + a.max_new_tokens + 64

    eng = Engine(cfg, device=local, max_batch=B, max_seq_len=max_seq, max_patches=pv.shape[0] + 64,
                 max_prefill_tokens=int(lens.sum()) + 64)
    eng.load_state_dict(sd)
    pix_dev = eng.to_device(pv)                      # inputs resident in HBM before the timed region
    setup_s = time.perf_counter() - t_setup

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    def step():
        return eng.generate(ids, lens, pix_dev, grid, a.max_new_tokens, (), pixel_on_device=True)

    for _ in range(a.warmup):
        step()
    eng.synchronize(); torch.cuda.synchronize(); barrier()
    t0 = time.perf_counter()
    phase = {"vit_ms": 0.0, "prefill_ms": 0.0, "decode_ms": 0.0, "vit_attn_ms": 0.0}
    last = None
    for _ in range(a.steps):
        out, out_lens = step()
        st = eng.stats()                             # device-side HIP-event times of this step
        for k in phase:
            phase[k] += st[k]
        last = st
    eng.synchronize(); torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    # the only data-path collective: gather the generated token ids on rank 0
    gathered = dp.gather_token_ids(out, out_lens, page_index=[rank * B + i for i in range(B)])
    if use_dist:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        K = a.steps
        pages_total = world * B * K
        new_tok = int(out_lens.sum()) * world * K
        attn_s = phase["vit_attn_ms"] / 1e3
        attn_tflops = last["vit_attn_flops"] * K / attn_s / 1e12 if attn_s > 0 else 0.0
        dec_s = phase["decode_ms"] / 1e3
        dec_gbs = last["decode_bytes"] * K / dec_s / 1e9 if dec_s > 0 else 0.0
        vit_s = phase["vit_ms"] / 1e3
        traffic = None
        tf = ROOT / "profiles" / "r01_flash_attn_traffic.json"
        if a.workload == "a4" and B == 8 and tf.exists():
            # HBM-side bytes per launch of the roofline kernel, from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
            # over this same command (corrected as MI355X_MICROARCH.md §HBM prescribes); see the file for the method.
            traffic = json.loads(tf.read_text())["traffic_bytes_per_launch"]
        res = {
            "metric": f"pages/sec, dots.ocr 1.7B bf16, A4@200dpi page batch (ViT + prefill + {a.max_new_tokens}-token greedy decode)",
            "value": pages_total / dt, "unit": "pages/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic pages (PIL text lines), seeded random weights at the checkpoint's dimensions",
            "config": {"workload": f"{a.workload}: {B} pages/GPU of {size[0]}x{size[1]} px -> {int(pv.shape[0] // B)} patches, "
                                   f"{int(lens[0])} prompt tokens/page, max_new_tokens={a.max_new_tokens}, EOS disabled",
                       "pages_per_gpu": B, "parallelism": f"dp{world}"},
            "output_tok_s": new_tok / dt,
            "decode_tok_s": int(out_lens.sum()) * K / dec_s if dec_s > 0 else None,
            "phase_ms_per_step": {k: v / K for k, v in phase.items()},
            "roofline": {"bound": "mfma", "kernel": "flash_attn_kernel<false> (ViT bidirectional var-len attention)",
                         "achieved": attn_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": attn_tflops / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch (PMC, profiles/r01_flash_attn_traffic.json)",
                         "algorithmic_flops_per_launch": last["vit_attn_flops"] / max(1, last["vit_attn_launches"]),
                         "launches_per_step": last["vit_attn_launches"],
                         "avg_launch_ms": phase["vit_attn_ms"] / K / max(1, last["vit_attn_launches"])},
            "roofline_vit": {"bound": "mfma", "achieved": last["vit_flops"] * K / vit_s / 1e12 if vit_s > 0 else 0.0,
                             "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                             "frac": (last["vit_flops"] * K / vit_s / 1e12 / PEAK_BF16_TFLOPS) if vit_s > 0 else 0.0},
            "roofline_decode": {"bound": "hbm", "achieved": dec_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": dec_gbs / PEAK_HBM_GBS, "ms_per_decode_step": phase["decode_ms"] / K / max(1, last["decode_steps"])},
            "gathered_pages": len(gathered), "setup_s": setup_s,
        }
        if world == 1 and not a.no_cpu_baseline:
            cores = min(os.cpu_count() or 1, 64)
            res["cpu_baseline"] = cpu_baseline(cfg, sd, cores)
        print(json.dumps(res), flush=True)
    eng.close()
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
