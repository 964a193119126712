"""Grid-barrier cost / visibility sweep on the GPU box (see csrc/probe_sync.hip).  Prints one JSON line per case."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dots_ocr_amd import _lib  # noqa: E402

lib = _lib.load()
lib.dots_probe_grid_barrier.restype = C.c_int32
lib.dots_probe_grid_barrier.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_int32)]
NB = 1000
for n_wg, lds in ((256, 0), (256, 80 * 1024), (512, 0), (128, 0)):
    for mode in (0, 1, 2, 3):
        best = None
        for rep in range(3):
            ms, st = C.c_float(0), (C.c_int32 * 2)(0, 0)
            rc = lib.dots_probe_grid_barrier(n_wg, 256, NB, mode, lds, C.byref(ms), st)
            r = {"n_wg": n_wg, "lds": lds, "mode": mode, "rc": rc, "us_per_barrier": round(ms.value * 1e3 / NB, 3),
                 "stale_reads": st[0], "timeouts": st[1]}
            if best is None or r["us_per_barrier"] < best["us_per_barrier"] or r["stale_reads"] or r["timeouts"]:
                best = r if not (best and (best["stale_reads"] or best["timeouts"])) else best
        print(json.dumps(best), flush=True)
