"""Page-batch data parallelism.

Pages are independent units (reference dots_ocr/parser.py:265-277 builds independent per-page
tasks and re-sorts results by page_no at :292), the 3 B-parameter model fits one MI355X hundreds
of times over, so the path shards by PAGE: one process per GPU, a full replica each, no
collective anywhere on the data path except ONE gather of the generated token ids at the end
(KB-scale, latency-bound — RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np


def page_cost(n_patches: int, expected_new_tokens: int = 1024) -> float:
    """Relative cost of one page, known after smart_resize and before any GPU work: ViT attention is
    quadratic in patches, the dense layers linear, decode linear in output tokens (SURVEY §8(d) formulas,
    in units of TFLOP-equivalents at the measured phase rates)."""
    n = float(n_patches)
    vit = 42 * (4 * n * n * 1536 + 2 * n * (4 * 1536 ** 2 + 3 * 1536 * 4224)) / 1e12
    prefill = 2 * 1.31e9 * (n / 4 + 250) / 1e12
    decode = expected_new_tokens * 0.004
    return vit + prefill + decode


def shard_pages(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first bin packing: page indices per rank, each rank's list in input order.
    Mixed batches vary 40x in cost (360 vs 14 400 vision tokens), round-robin would leave ranks idle."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    bins: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += costs[i]
    return [sorted(b) for b in bins]


class PageQueue:
    """Shared work queue over the pages of ONE job, for jobs much larger than the ranks' slots (SURVEY §8(e) "scaling risks": with a
    static shard the ranks that drew long outputs finish last while the others idle).  The job's pages are put in ONE order known to
    every rank (costliest first: big pages start early, small ones fill the tail); a rank takes the next page(s) whenever its slots
    drain.  The only shared state is one integer, fetched-and-added on the process group's HOST-side key-value store (the TCPStore
    torch.distributed already runs for the rendezvous): no GPU collective, nothing on the data path — the result gather stays the
    job's only collective.  Single process (no store): a local counter, i.e. the same order served to one rank.
    Opt-in (bench.py --page-queue, ContinuousBatcher.run_pull); the default stays the static LPT shard of shard_pages."""

    def __init__(self, costs: Sequence[float], store=None, key: str = "dots_ocr/page_queue", world_size: int = 1):
        self.order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
        self.store, self.key, self._local = store, key, 0
        # one call hands out at most half a rank's fair share: a rank with many free slots (32 per GPU in bench.py's mixed64) fills them over
        # several scheduler steps instead of emptying the queue before the other ranks have asked
        self.max_take = max(1, len(self.order) // (2 * max(1, int(world_size))))

    @staticmethod
    def default_store():
        """the initialised default process group's store, or None (single process)"""
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from torch.distributed import distributed_c10d as c10d
                return c10d._get_default_store()
        except Exception:
            pass
        return None

    def take(self, n: int = 1) -> List[int]:
        """the next n page indices of the job (fewer, or none, at the end); each index is handed out exactly once over all ranks"""
        n = max(1, min(int(n), self.max_take))
        if self.store is None:
            start, self._local = self._local, self._local + n
        else:
            start = int(self.store.add(self.key, n)) - n             # atomic fetch-and-add on the store's server
        return self.order[start:start + n] if start < len(self.order) else []


def gather_token_ids(out_ids: np.ndarray, out_lens: np.ndarray, page_index: Sequence[int] | None = None):
    """All ranks contribute (out_ids [b, n], out_lens [b]); returns on every rank the list of
    (page_index, token list) for the whole job, sorted by page index.  Single-process: local only.
    Two collectives: all_gather of the per-rank (count, max_len), then one padded all_gather of ids."""
    import torch
    b = int(out_lens.shape[0])
    local_idx = list(page_index) if page_index is not None else None
    try:
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized()
    except Exception:
        active = False
    if not active:
        idx = local_idx if local_idx is not None else list(range(b))
        return sorted((int(idx[i]), out_ids[i, : out_lens[i]].tolist()) for i in range(b))
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    width = int(out_lens.max()) if b else 0
    meta = torch.tensor([b, width], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    bmax = max(int(m[0]) for m in metas)
    wmax = max(int(m[1]) for m in metas)
    # row = [page_index, length, ids...]
    buf = torch.full((bmax, wmax + 2), -1, dtype=torch.int64)
    for i in range(b):
        gi = local_idx[i] if local_idx is not None else rank * bmax + i
        buf[i, 0], buf[i, 1] = gi, int(out_lens[i])
        buf[i, 2: 2 + int(out_lens[i])] = torch.from_numpy(out_ids[i, : out_lens[i]].astype(np.int64))
    buf = buf.to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    res = []
    for t in bufs:
        t = t.cpu()
        for row in t:
            if int(row[0]) >= 0:
                res.append((int(row[0]), row[2: 2 + int(row[1])].tolist()))
    return sorted(res)
