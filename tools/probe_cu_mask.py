"""Which CUs does a CU-masked stream use?  (GPU box helper; see csrc/probe_sync.hip)"""
import ctypes as C
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dots_ocr_amd import _lib  # noqa: E402

lib = _lib.load()
lib.dots_probe_cu_mask.restype = C.c_int32
lib.dots_probe_cu_mask.argtypes = [C.POINTER(C.c_uint32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]


def run(name, bits):
    n_wg = 2048
    out = np.zeros((n_wg, 2), np.uint32)
    if bits is None:
        rc = lib.dots_probe_cu_mask(None, 0, n_wg, 256, 80 * 1024, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    else:
        words = np.zeros(8, np.uint32)
        for b in bits:
            words[b // 32] |= np.uint32(1 << (b % 32))
        rc = lib.dots_probe_cu_mask(words.ctypes.data_as(C.POINTER(C.c_uint32)), 8, n_wg, 256, 80 * 1024, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    hw, xcc = out[:, 0], out[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    per = collections.Counter()
    for t in set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())):
        per[t[0]] += 1
    print(f"{name}: rc={rc} distinct CUs {sum(per.values())} per XCC {[per.get(i, 0) for i in range(8)]}", flush=True)


run("all (null stream)", None)
run("bits 0..255", range(256))
run("bits 0..31", range(32))
run("bits 0..63", range(64))
run("bits 224..255", range(224, 256))
run("bits i%8==0", [i for i in range(256) if i % 8 == 0])
run("bits 0..207", range(208))
run("bits 208..255", range(208, 256))
