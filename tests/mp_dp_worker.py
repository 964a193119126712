"""Worker of tests/test_multi_gpu.py: ONE rank of a page-batch data-parallel job on the real engine, launched by
`python -m torch.distributed.run --nproc-per-node N tests/mp_dp_worker.py OUT.json` (exactly how the driver launches bench.py).

Job: 10 mixed-size pages of the small-dims model (tiny dots.ocr, seeded random weights), cost-sharded over the ranks
(dp.shard_pages), each shard continuously batched over 3 slots (scheduler.ContinuousBatcher), token ids gathered on every rank
(dp.gather_token_ids: the only collective of the data path).  Rank 0 writes {"world", "rccl_ranks", "backend", "gathered"} to OUT.json.
Reference call sites: dots_ocr/parser.py:265-277 (independent per-page tasks), :292 (results re-sorted by page number).

DOTS_TEST_DP_BACKEND=gloo + DOTS_TEST_DP_ONE_GPU=1 run every rank on cuda:0 with the gloo backend (1-GPU boxes: everything but RCCL).
"""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402
import torch  # noqa: E402

GRIDS = [(1, 4, 6), (1, 8, 8), (1, 4, 4), (1, 10, 6), (1, 6, 6), (1, 4, 8), (1, 12, 8), (1, 4, 4), (1, 6, 4), (1, 8, 6)]
CAPS = [24, 40, 7, 33, 16, 28, 12, 40, 9, 21]


def job(cfg):
    """(input_ids, pixel_values, grid_thw, cap) per page — a pure function of the page index."""
    pages = []
    for i, (g, cap) in enumerate(zip(GRIDS, CAPS)):
        gen = torch.Generator().manual_seed(900 + i)
        n = g[1] * g[2]
        pv = torch.randn(n, cfg.vision.patch_dim, generator=gen).numpy()
        ids = torch.cat([torch.randint(0, cfg.vocab_size - 8, (3,), generator=gen), torch.full((n // 4,), cfg.image_token_id),
                         torch.randint(0, cfg.vocab_size - 8, (4 + i % 3,), generator=gen)]).numpy().astype(np.int32)
        pages.append((ids, pv, np.asarray([g], np.int64), cap))
    return pages


def run_shard(eng, pages, mine):
    from dots_ocr_amd.scheduler import ContinuousBatcher, Request
    outs = ContinuousBatcher(eng, eos_ids=(), chunk=8).run([Request(*pages[i]) for i in mine]) if mine else []
    width = max([len(o) for o in outs], default=1)
    ids = np.zeros((len(outs), width), np.int32)
    lens = np.zeros((len(outs),), np.int32)
    for j, o in enumerate(outs):
        ids[j, :len(o)], lens[j] = o, len(o)
    return ids, lens


def make_engine(device):
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.weights import random_state_dict
    cfg = DotsConfig.tiny(layers=2, v_layers=2, vocab=1024)
    eng = Engine(cfg, device=device, max_batch=3, max_seq_len=256, max_patches=1024, max_prefill_tokens=512)
    eng.load_state_dict(random_state_dict(cfg, seed=21))
    return cfg, eng


def main():
    import torch.distributed as dist
    from dots_ocr_amd import dp
    out_path = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("DOTS_TEST_DP_BACKEND", "nccl")
    device = 0 if os.environ.get("DOTS_TEST_DP_ONE_GPU") else local
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: no interface / hostname discovery (the box's hostname may not resolve: tens of seconds of time-outs)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend)
    cfg, eng = make_engine(device)
    pages = job(cfg)
    shards = dp.shard_pages([dp.page_cost(p[1].shape[0], p[3]) for p in pages], world)
    ids, lens = run_shard(eng, pages, shards[rank])
    gathered = dp.gather_token_ids(ids, lens, page_index=shards[rank])
    ones = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    if rank == 0:
        Path(out_path).write_text(json.dumps({"world": world, "rccl_ranks": int(ones.item()), "backend": dist.get_backend(),
                                              "pages_per_rank": [len(s) for s in shards], "gathered": gathered}))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
