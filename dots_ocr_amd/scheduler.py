"""Continuous batching over the engine's sequence slots.

The reference gets this from its vLLM server: ``DotsOCRParser`` fires one request per page from a thread pool
(dots_ocr/parser.py:138-166, model/inference.py:8-50) and the server keeps its running batch full, admitting a new
page whenever one finishes.  Here the same policy runs in-process on top of the C-ABI slot calls
(include/dots_ocr_hip.h "Continuous batching"): FIFO admission into free slots within the ViT / prefill workspace
budgets, fixed-size decode chunks replayed from a captured graph, finished sequences read out and their slots reused.

A page's tokens do not depend on what shares the batch with it (every kernel on the path is row-independent), so the
result of a request is the same as a single-sequence ``generate`` — tests/test_model_gpu.py checks exactly that.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass
from typing import Deque, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Request:
    input_ids: np.ndarray                       # int32 [T]; image pad tokens already expanded (processor output)
    pixel_values: Optional[object] = None       # float32 [n_patches, patch_dim]: numpy array or torch CUDA tensor
    grid_thw: Optional[np.ndarray] = None       # int64 [n_img, 3]
    max_new_tokens: int = 128
    tag: object = None

    def n_patches(self) -> int:
        if self.grid_thw is None:
            return 0
        g = np.asarray(self.grid_thw, dtype=np.int64).reshape(-1, 3)
        return int((g[:, 0] * g[:, 1] * g[:, 2]).sum())


class ContinuousBatcher:
    """submit() requests at any time, call step() in a loop (or run() for a closed set)."""

    def __init__(self, engine, eos_ids: Sequence[int] = (), chunk: int = 16, headroom_pages: Optional[int] = 2, prefetch: int = 0,
                 tower_steps_per_page: int = 48, admit_group: Optional[int] = None):
        self.engine = engine
        self.chunk = max(1, int(chunk))
        # free KV pages kept per running sequence when admitting (see plan_admission).  None = FULL RESERVATION: a sequence is admitted only
        # when the pool could hold prompt + max_new_tokens of every running sequence and of this one, so nothing is ever truncated by a dry
        # pool (the policy of rounds 1-2; fewer sequences in flight when caps are generous, as with the reference's max_new_tokens=24000).
        self.full_reservation = headroom_pages is None
        self.headroom_pages = 0 if headroom_pages is None else max(0, int(headroom_pages))
        # look-ahead: the vision tower of the next `prefetch` queued requests runs on the engine's CU-masked side stream while the
        # occupied slots keep decoding on the other CU partition (Engine.vit_prefetch); that group is then admitted as a whole — prefill
        # only, no tower in the decode loop's way — as soon as it has the slots and pages.  1 suits requests that finish at different
        # times (a slot is refilled as soon as it frees), n_slots suits closed sets of equal length.  0 = off.
        self.prefetch = max(0, int(prefetch)) if hasattr(engine, "vit_prefetch") else 0
        # decode steps that pass while ONE page's tower runs on the side partition (A4 on 160-192 CUs: ~150 ms against 2.6-3.3 ms per step)
        self.tower_steps_per_page = max(1, int(tower_steps_per_page))
        # at most this many requests per ORDINARY admission (None: as many as fit).  With look-ahead on, the ordinary path only runs when
        # nothing is prefetched — the cold start of a queue: admitting `prefetch`-sized groups there too starts the decode loop after ONE
        # small tower instead of after the towers of every free slot, and the later groups' towers run beside it (a closed set of equal
        # caps then also finishes staggered, one group per tower, instead of all at once with nothing left to overlap)
        self.admit_group = None if admit_group is None else max(1, int(admit_group))
        self._ahead: List[Tuple[int, Request]] = []             # requests whose tower has been prefetched, in packed order
        self._ahead_keep = None                                  # their (device) pixels stay alive until the rows are taken
        self.n_slots = int(engine.max_batch)
        self.max_patches = int(engine.max_patches)
        self.max_prefill_tokens = int(engine.max_prefill_tokens)
        self.max_seq_len = int(engine.max_seq_len)
        self.pending: Deque[Tuple[int, Request]] = deque()
        self.running: Dict[int, Tuple[int, Request]] = {}        # slot -> (request id, request)
        self._next_id = 0
        self.decode_steps = 0
        self.admissions = 0
        self._last_lens: Dict[int, int] = {}                     # slot -> tokens generated as of the last poll
        self.kv_truncated = 0                                    # sequences ended early by a dry KV pool (finish reason "kv_pool_exhausted")
        engine.set_eos(list(eos_ids))
        if hasattr(engine, "slots_reset"):           # start from an empty engine: no occupied slot, every KV page in the pool
            engine.slots_reset()                     # (a static batch or a failed run may have left both behind)
        else:
            fin, _ = engine.slots_poll()
            for s in range(self.n_slots):
                if fin[s] >= 0:
                    engine.slot_release(s)

    # ------------------------------------------------------------------ queue
    def submit(self, req: Request) -> int:
        ids = np.ascontiguousarray(req.input_ids, dtype=np.int32).reshape(-1)
        if ids.shape[0] < 1 or ids.shape[0] >= self.max_seq_len:
            raise ValueError(f"prompt of {ids.shape[0]} tokens does not fit max_seq_len={self.max_seq_len}")
        if ids.shape[0] > self.max_prefill_tokens:
            raise ValueError(f"prompt of {ids.shape[0]} tokens exceeds max_prefill_tokens={self.max_prefill_tokens}")
        if req.n_patches() > self.max_patches:
            raise ValueError(f"request has {req.n_patches()} vision patches, more than max_patches={self.max_patches}")
        req.input_ids = ids
        # like HF generate, stop at the context capacity instead of failing
        req.max_new_tokens = max(1, min(int(req.max_new_tokens), self.max_seq_len - ids.shape[0]))
        if hasattr(self.engine, "kv_pool_info") and self._admit_pages(ids.shape[0], req.max_new_tokens) > self.engine.kv_pool_info()[0]:
            raise ValueError(f"prompt of {ids.shape[0]} tokens needs {self._admit_pages(ids.shape[0], req.max_new_tokens)} KV pages, "
                             f"the pool holds {self.engine.kv_pool_info()[0]}")
        if self.full_reservation and hasattr(self.engine, "kv_pool_info") and self._worst_pages(req) > self.engine.kv_pool_info()[0]:
            raise ValueError(f"prompt + max_new_tokens = {ids.shape[0]} + {req.max_new_tokens} tokens need {self._worst_pages(req)} KV pages under full "
                             f"reservation, the pool holds {self.engine.kv_pool_info()[0]}")
        rid = self._next_id
        self._next_id += 1
        self.pending.append((rid, req))
        return rid

    @property
    def idle(self) -> bool:
        return not self.pending and not self.running and not self._ahead

    def free_slots(self) -> List[int]:
        return [s for s in range(self.n_slots) if s not in self.running]

    # ------------------------------------------------------------------ admission
    ADMIT_AHEAD = 64                                 # tokens beyond the prompt dots_slots_prefill reserves (engine.hip KV_ADMIT_AHEAD)

    def _admit_pages(self, prompt: int, max_new: int) -> int:
        return (min(prompt + min(int(max_new), self.ADMIT_AHEAD), self.max_seq_len) + 63) // 64

    def _worst_pages(self, r: Request) -> int:
        return (min(int(r.input_ids.shape[0]) + int(r.max_new_tokens), self.max_seq_len) + 63) // 64

    def _fits_full_reservation(self, admitted: Sequence[Request], new: Sequence[Request]) -> bool:
        """Full reservation (headroom_pages=None): the worst case of everything in flight — running, already in this group, prefetched —
        plus `new` must fit the pool.  Applied on EVERY admission path (ordinary, look-ahead, prefetched group)."""
        if not self.full_reservation or not hasattr(self.engine, "kv_pool_info"):
            return True
        total = self.engine.kv_pool_info()[0]
        committed = sum(self._worst_pages(r) for _, r in self.running.values()) + sum(self._worst_pages(r) for r in admitted)
        return committed + sum(self._worst_pages(r) for r in new) <= total

    def plan_admission(self) -> List[Tuple[int, int, Request]]:
        """FIFO: pop requests while a slot is free and the group fits the ViT and prefill workspaces.
        Lowest slots first, so the decode graph covers as few rows as possible."""
        free = self.free_slots()
        group, patches, tokens = [], 0, 0
        pages_free = self.engine.kv_pool_info()[1] if hasattr(self.engine, "kv_pool_info") else 1 << 30
        while self.pending and free and (self.admit_group is None or len(group) < self.admit_group):
            rid, req = self.pending[0]
            p, t = req.n_patches(), int(req.input_ids.shape[0])
            if group and (patches + p > self.max_patches or tokens + t > self.max_prefill_tokens):
                break
            # paged KV, on demand: a sequence reserves its prompt + ADMIT_AHEAD tokens now and grows page by page; admit only while every
            # running sequence (and this one) could still take `headroom` more pages, so that a dry pool — which ends a sequence early
            # at what its pages hold — stays the exception.  Nothing running: admit whatever fits (submit() checked that it can).
            need = self._admit_pages(t, req.max_new_tokens)
            reserve = self.headroom_pages * (len(self.running) + len(group) + 1) if (self.running or group) else 0
            if (self.running or group) and not self._fits_full_reservation([r for _, _, r in group], [req]):
                break                                # worst case of everything in flight must fit the pool (submit() checked a lone request)
            if need + reserve > pages_free:
                break                                                                    # wait for a running sequence to return its pages
            pages_free -= need
            self.pending.popleft()
            group.append((free.pop(0), rid, req))
            patches += p
            tokens += t
        return group

    def _pixels(self, with_img):
        """(pixel values, grid, on_device, keep-alive) of the requests' images, packed in order"""
        grid = np.concatenate([np.asarray(r.grid_thw, dtype=np.int64).reshape(-1, 3) for _, r in with_img])
        pvs = [r.pixel_values for _, r in with_img]
        if all(hasattr(p, "is_cuda") and p.is_cuda for p in pvs):
            import torch
            pv = pvs[0] if len(pvs) == 1 else torch.cat(pvs, dim=0)
            pv = pv.contiguous().float()
            torch.cuda.synchronize(pv.device)
            return pv.data_ptr(), grid, True, pv
        host = [p.detach().cpu().numpy() if hasattr(p, "detach") else np.asarray(p) for p in pvs]
        return np.concatenate(host).astype(np.float32, copy=False), grid, False, None

    def _prefill(self, group):
        # image rows are consumed in packed order, so sequences with images keep their relative order: pack the group as is
        slots = [s for s, _, _ in group]
        lens = [int(r.input_ids.shape[0]) for _, _, r in group]
        caps = [int(r.max_new_tokens) for _, _, r in group]
        self.engine.slots_prefill(slots, np.concatenate([r.input_ids for _, _, r in group]), lens, caps)
        for s, rid, r in group:
            self.running[s] = (rid, r)
            self._last_lens[s] = 0                   # a reused slot must not inherit its previous occupant's length (_soon_free counted a
                                                     # just-admitted long request as "about to finish" and the look-ahead group was sized for slots that stayed busy)
        self.admissions += 1

    def _admit(self, group):
        with_img = [(rid, r) for _, rid, r in group if r.n_patches() > 0]
        if with_img:
            pv, grid, on_dev, keep = self._pixels(with_img)
            self.engine.vit_forward(pv, grid, on_device=on_dev)
            if on_dev:
                self.engine.synchronize()            # `keep` may be a temporary
        self._prefill(group)

    # ------------------------------------------------------------------ look-ahead (prefetched towers)
    def _admit_ahead(self) -> bool:
        """The group whose tower was prefetched goes in as a whole as soon as it has the slots and the pages."""
        free = self.free_slots()
        if len(free) < len(self._ahead):
            return False
        # sequences are decoding and the tower is still running: taking the group now would queue every decode chunk behind the tower
        # (vit_take makes the engine's stream wait for it) — keep decoding, take it when its rows exist.  Nothing running: wait for it.
        if self.running and hasattr(self.engine, "vit_ready") and not self.engine.vit_ready():
            return False
        if hasattr(self.engine, "kv_pool_info"):
            need = sum(self._admit_pages(int(r.input_ids.shape[0]), r.max_new_tokens) for _, r in self._ahead)
            reserve = self.headroom_pages * (len(self.running) + len(self._ahead)) if self.running else 0
            if need + reserve > self.engine.kv_pool_info()[1]:
                return False
            if self.running and not self._fits_full_reservation([], [r for _, r in self._ahead]):
                return False
        group = self._ahead
        try:
            self.engine.vit_take()
            self._prefill([(free[i], rid, r) for i, (rid, r) in enumerate(group)])
        except Exception:
            # the rows are gone (or were never there): the group goes back to the head of the queue so that the caller's failure
            # handling — which walks `pending` and `running` — sees its requests, and the next step does not take a stale batch
            self._ahead, self._ahead_keep = [], None
            self.pending.extendleft(reversed(group))
            raise
        self._ahead, self._ahead_keep = [], None
        return True

    def _look_ahead(self):
        """Start the tower of the next queued requests (FIFO prefix, image requests only) on the side stream."""
        if not self.prefetch or self._ahead or not self.pending:
            return
        total_pages = self.engine.kv_pool_info()[0] if hasattr(self.engine, "kv_pool_info") else 1 << 30
        group, patches, tokens, pages = [], 0, 0, 0
        # A prefetched group goes in only as a whole, and nothing may overtake it: a group larger than the slots that are free — or will be
        # by the time its tower is done — would hold freed slots idle while it waits for the rest; with uneven output lengths (the
        # reference's max_new_tokens=24000) for a long time.  The tower of k pages lasts about k * tower_steps_per_page decode
        # steps, so the group is the LARGEST k <= prefetch for which k slots are free or within that horizon (+ two chunks) of their caps:
        # equal caps (a document's pages: everything finishes together) prefetch the whole next group while the current one decodes — the
        # round-5 rule "free now or within two chunks" started ONE tower there and ran the other 31 at admission, nothing beside them
        # (mixed64: 4.40 -> 4.00 pages/s, profiles/r05_mixed64_lookahead_regression.txt).  At least one, so that a full engine still
        # hides the next tower.
        cap = 1
        for k in range(min(self.prefetch, self.n_slots), 1, -1):
            if len(self.free_slots()) + self._soon_free(2 * self.chunk + k * self.tower_steps_per_page) >= k:
                cap = k
                break
        while self.pending and len(group) < cap:
            rid, req = self.pending[0]
            p, t = req.n_patches(), int(req.input_ids.shape[0])
            pg = self._admit_pages(t, req.max_new_tokens)
            if p == 0 or patches + p > self.max_patches or tokens + t > self.max_prefill_tokens or pages + pg > total_pages:
                break                                # a text-only request goes through the ordinary admission, in its turn
            if self.full_reservation and sum(self._worst_pages(r) for _, r in group) + self._worst_pages(req) > total_pages:
                break                                # full reservation: the group must fit an EMPTY pool by its worst case (its tower may run ahead;
                                                     # _admit_ahead holds it back until it also fits beside what is still running)
            self.pending.popleft()
            group.append((rid, req))
            patches, tokens, pages = patches + p, tokens + t, pages + pg
        if not group:
            return
        try:
            pv, grid, on_dev, keep = self._pixels(group)
            self.engine.vit_prefetch(pv, grid, on_device=on_dev)
        except Exception:
            self.pending.extendleft(reversed(group))     # nothing was prefetched: the requests keep their place in the queue
            raise
        self._ahead, self._ahead_keep = group, keep

    def _soon_free(self, horizon: Optional[int] = None) -> int:
        """running sequences within `horizon` decode steps (default: two chunks) of their generation cap (lengths as of the last poll)"""
        horizon = 2 * self.chunk if horizon is None else horizon
        return sum(1 for s, (_, r) in self.running.items() if self._last_lens.get(s, 0) + horizon >= int(r.max_new_tokens))

    # ------------------------------------------------------------------ main loop
    def _collect(self) -> List[Tuple[int, Request, np.ndarray]]:
        fin, lens = self.engine.slots_poll()
        done = []
        for s in sorted(self.running):
            if fin[s] == 1:
                rid, req = self.running.pop(s)
                toks = self.engine.slot_read(s, int(lens[s]))
                # a sequence the engine ended early because the KV pool ran dry (its cap was lowered to what its pages hold) is not an
                # ordinary "length" stop: say so on the request, count it, let the server report it
                req.kv_truncated = False
                if hasattr(self.engine, "slot_capacity"):
                    _, limit = self.engine.slot_capacity(s)
                    req.kv_truncated = bool(limit < int(req.input_ids.shape[0]) + int(req.max_new_tokens) and len(toks) >= limit - int(req.input_ids.shape[0]))
                    self.kv_truncated += int(req.kv_truncated)
                done.append((rid, req, toks))
                self.engine.slot_release(s)
        self._last_lens = {s: int(lens[s]) for s in self.running}      # after the finished slots have left: only what is still decoding
        return done

    def step(self) -> List[Tuple[int, Request, np.ndarray]]:
        """Admit what fits, run one decode chunk, return the requests that finished: (id, request, new token ids)."""
        if self._ahead:                              # a prefetched group waits for its slots: nothing may overtake it
            admitted = self._admit_ahead()
        else:
            group = self.plan_admission()
            admitted = bool(group)
            if group:
                try:
                    self._admit(group)
                except Exception:
                    # vit_forward / slots_prefill failed: plan_admission popped these requests and nothing registered them yet — put them
                    # back at the head of the queue (in order) so that the caller's failure handling, which walks `pending`, `running` and
                    # the prefetched group, finds every request (server.py fails their futures; a lost request would hang its HTTP call)
                    for s, _, _ in group:
                        self.running.pop(s, None)
                    self.pending.extendleft(reversed([(rid, r) for _, rid, r in group]))
                    raise
        self._look_ahead()
        if admitted:
            done = self._collect()                   # a 1-token cap or an immediate EOS finishes at prefill
            if done:
                return done
        if not self.running:
            if self._ahead:                          # nothing runs, so every slot and page is free: the group fits by construction
                raise RuntimeError("a prefetched group could not be admitted into an empty engine")
            if self.pending:                         # nothing runs, nothing could be admitted: it never will be
                rid, req = self.pending[0]           # left in the queue: whoever handles the error finds (and fails) it there
                raise RuntimeError(f"request {rid} ({req.input_ids.shape[0]} prompt tokens, {req.n_patches()} patches) cannot be admitted "
                                   f"into an empty engine (KV pool {self.engine.kv_pool_info() if hasattr(self.engine, 'kv_pool_info') else '?'})")
            return []
        self.engine.slots_decode(self.chunk)
        self.decode_steps += self.chunk
        return self._collect()

    def run_pull(self, pull, low_water: Optional[int] = None) -> Dict[object, np.ndarray]:
        """An open-ended job: `pull(k)` returns up to k (key, Request) pairs from a source shared with other ranks ([] = the source is dry;
        dots_ocr_amd.dp.PageQueue).  Requests are pulled only as they can be used — while fewer than `low_water` (default: the look-ahead
        group size, at least one) are queued beyond what the free slots can take — so a rank never hoards pages another rank could start.
        Returns {key: new token ids}."""
        low = max(1, self.prefetch) if low_water is None else max(1, int(low_water))
        keys: Dict[int, object] = {}
        out: Dict[object, np.ndarray] = {}
        dry = False
        while True:
            want = low + len(self.free_slots()) - len(self._ahead) - len(self.pending)       # what the free slots can take now + the look-ahead's next group
            if not dry and want > 0:
                got = pull(want)
                dry = not got
                for key, req in got:
                    keys[self.submit(req)] = key
            if self.idle:
                if dry:
                    return out
                continue
            for rid, _, toks in self.step():
                out[keys.pop(rid)] = toks

    def run(self, requests: Iterable[Request]) -> List[np.ndarray]:
        ids = [self.submit(r) for r in requests]
        out: Dict[int, np.ndarray] = {}
        while not self.idle:
            for rid, _, toks in self.step():
                out[rid] = toks
        return [out[i] for i in ids]
