"""CPU stand-ins for the engine's slot calls (scheduler / server logic tests; no numerics)."""
import numpy as np


class FakeSlotEngine:
    """Same slot surface as dots_ocr_amd.engine.Engine.  A sequence's tokens come from `script(prompt) -> list[int]`;
    it stops at its cap or at the first EOS id, like the device bookkeeping does."""

    def __init__(self, script, max_batch=3, max_patches=100, max_prefill_tokens=64, max_seq_len=128):
        self.script = script
        self.max_batch, self.max_patches, self.max_prefill_tokens, self.max_seq_len = max_batch, max_patches, max_prefill_tokens, max_seq_len
        self.slots = {}            # slot -> dict(prompt, cap, plan, out, done)
        self.log = []
        self.eos = []
        self.sampling = (0.0, 1.0, 0)

    def set_eos(self, eos):
        self.eos = list(eos)

    def set_sampling(self, t, p, seed):
        self.sampling = (t, p, seed)

    def synchronize(self):
        pass

    def vit_forward(self, pv, grid, on_device=False):
        self.log.append(("vit", int(np.asarray(grid)[:, 1:].prod(axis=1).sum()), len(pv)))

    def vit_prefetch(self, pv, grid, on_device=False, after_prefill=False):
        assert not getattr(self, "_pref", None), "a prefetched batch is waiting"
        self._pref = int(np.asarray(grid)[:, 1:].prod(axis=1).sum())
        self.log.append(("prefetch", self._pref, len(pv)))

    def vit_take(self):
        assert getattr(self, "_pref", None), "nothing prefetched"
        self.log.append(("take", self._pref))
        self._pref = None

    def _advance(self, st):
        if st["done"]:
            return
        tok = st["plan"][len(st["out"])] if len(st["out"]) < len(st["plan"]) else 0
        st["out"].append(int(tok))
        st["done"] = tok in self.eos or len(st["out"]) >= st["cap"]

    def slots_prefill(self, slots, ids, lens, caps):
        assert sum(lens) == len(ids) <= self.max_prefill_tokens
        off = 0
        for s, n, c in zip(slots, lens, caps):
            assert 0 <= s < self.max_batch and s not in self.slots
            prompt = np.asarray(ids[off:off + n]).copy()
            self.slots[s] = dict(prompt=prompt, cap=int(c), plan=list(self.script(prompt)), out=[], done=False)
            self._advance(self.slots[s])
            off += n
        self.log.append(("prefill", tuple(slots)))

    def slots_decode(self, n):
        assert self.slots
        for _ in range(n):
            for st in self.slots.values():
                self._advance(st)
        self.log.append(("decode", n, tuple(sorted(self.slots))))

    def slots_poll(self):
        fin = np.full(self.max_batch, -1, np.int32)
        lens = np.zeros(self.max_batch, np.int32)
        for s, st in self.slots.items():
            fin[s], lens[s] = int(st["done"]), len(st["out"])
        return fin, lens

    def slot_read(self, s, capacity):
        return np.asarray(self.slots[s]["out"][:capacity], dtype=np.int32)

    def slot_release(self, s):
        del self.slots[s]


class FakePagedEngine(FakeSlotEngine):
    """FakeSlotEngine + the paged KV pool of the real engine (engine.hip: reserve at admission, grow per decode chunk, lower the cap when
    the pool runs dry — pages of P positions allow P + 1 tokens in all)."""
    PAGE = 64

    def __init__(self, script, pool_pages, **kw):
        super().__init__(script, **kw)
        self.pool_pages, self.free = pool_pages, pool_pages
        self.capped = 0

    def kv_pool_info(self):
        return self.pool_pages, self.free

    def slots_reset(self):
        self.slots.clear()
        self.free = self.pool_pages

    def slots_prefill(self, slots, ids, lens, caps):
        need = [(min(n + min(c, 64), self.max_seq_len) + 63) // 64 for n, c in zip(lens, caps)]
        if sum(need) > self.free:
            raise RuntimeError("KV pool exhausted")
        super().slots_prefill(slots, ids, lens, caps)
        for s, n, c, p in zip(slots, lens, caps, need):
            self.slots[s].update(pages=p, limit=n + c, ctx=n)
            self.free -= p

    def slots_decode(self, n):
        for st in self.slots.values():
            if st["done"]:
                continue
            want = min(st["ctx"] + n, st["limit"] - 1)
            while st["pages"] * self.PAGE < want and self.free > 0:
                st["pages"] += 1
                self.free -= 1
            have = st["pages"] * self.PAGE
            if have < want:
                st["limit"] = have + 1
                st["cap"] = have + 1 - len(st["prompt"])
                self.capped += 1
                if len(st["out"]) >= st["cap"]:
                    st["done"] = True
            st["ctx"] = min(st["ctx"] + n, st["limit"] - 1)
        super().slots_decode(n)

    def slot_capacity(self, s):
        return self.slots[s]["pages"], self.slots[s]["limit"]

    def slot_release(self, s):
        self.free += self.slots[s]["pages"]
        super().slot_release(s)
