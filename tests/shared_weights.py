"""Full-size seeded random checkpoints shared by the GPU tests of one pytest session (round 6: five test modules built the same 6 GB state dict
and converted it to fp32 for the oracle each on its own — about 80 s of the suite's wall time on the GPU box).

    full_sd(seed)            the bf16 state dict of DotsConfig() (dots_ocr_amd.weights.random_state_dict); cached — DO NOT modify its tensors in place
    F32View(sd, base=None)   a read-only mapping over `sd` that converts a tensor to fp32 the first time the oracle asks for it and keeps it;
                             tensors that `sd` shares (same object) with the cached checkpoint reuse the session's conversion
"""
from __future__ import annotations

import os
from collections.abc import Mapping

_BF16 = {}
_F32 = {}          # id(bf16 tensor) -> fp32 tensor, for tensors of the cached checkpoints only


def full_sd(seed: int = 0):
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.weights import random_state_dict
    if seed not in _BF16:
        _BF16[seed] = random_state_dict(DotsConfig(), seed=seed, threads=min(32, os.cpu_count() or 8))
    return _BF16[seed]


class F32View(Mapping):
    def __init__(self, sd, skip_prefix: str | None = None):
        self._sd = sd if skip_prefix is None else {k: v for k, v in sd.items() if not k.startswith(skip_prefix)}
        self._own = {}
        self._shared = {id(v) for d in _BF16.values() for v in d.values()}      # (the cached dicts keep these tensors alive: ids stay unique)

    def __getitem__(self, k):
        v = self._sd[k]
        if id(v) in self._shared:
            if id(v) not in _F32:
                _F32[id(v)] = v.float()
            return _F32[id(v)]
        if k not in self._own:
            self._own[k] = v.float()
        return self._own[k]

    def __iter__(self):
        return iter(self._sd)

    def __len__(self):
        return len(self._sd)
