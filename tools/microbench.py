#!/usr/bin/env python3
"""Kernel microbenchmarks at the A4 workload's shapes, through the C ABI (GPU box only).
    python tools/microbench.py [gemm] [flash] [--iters N]
Times each op with torch CUDA events after syncing the engine stream; under rocprofv3 the kernel
trace / PMC counters give the per-dispatch numbers."""
import math
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dots_ocr_amd.config import DotsConfig  # noqa: E402
from dots_ocr_amd.engine import Engine  # noqa: E402

iters = 5
if "--iters" in sys.argv:
    iters = int(sys.argv[sys.argv.index("--iters") + 1])
what = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()] or ["gemm", "flash"]   # --iters N, --seqs N take numbers
eng = Engine(DotsConfig.tiny(), max_batch=2, max_seq_len=256, max_patches=256, max_prefill_tokens=256)


def timed(fn, flops, name):
    torch.cuda.synchronize()
    fn(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:50s} {dt * 1e3:9.3f} ms  {flops / dt / 1e12:8.1f} TFLOP/s", flush=True)


if "gemm" in what:
    M = 2 * 19824
    for (N, K, epi, nm) in [(4608, 1536, 0, "vit qkv"), (8448, 1536, 2, "vit fc1|fc3 swiglu"), (1536, 4224, 1, "vit fc2 +res"),
                            (1536, 1536, 1, "vit proj +res")]:
        A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        ldc = N // 2 if epi == 2 else N
        C = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16)
        timed(lambda: eng.op_gemm(A.data_ptr(), W.data_ptr(), 0, C.data_ptr() if epi == 1 else 0, C.data_ptr(), M, N, K, epi),
              2.0 * M * N * K, f"gemm {nm} M={M} N={N} K={K}")

if "flash" in what:
    H, n = 12, 19824
    seqs = int(sys.argv[sys.argv.index("--seqs") + 1]) if "--seqs" in sys.argv else 2
    lens = [n] * seqs
    T = sum(lens)
    Tpad = sum((x + 63) // 64 * 64 for x in lens)
    q = torch.randn(H, T, 128, device="cuda").bfloat16()
    k = torch.randn(H, T + 64, 128, device="cuda").bfloat16()
    vt = torch.randn(H, 128, Tpad, device="cuda").bfloat16()
    out = torch.zeros(T, H * 128, device="cuda", dtype=torch.bfloat16)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    timed(lambda: eng.op_flash_attn(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), cu, H, H, False, 1 / math.sqrt(128)),
          sum(4.0 * x * x * 128 * H for x in lens), f"flash attn bidirectional H={H} n={n} x{len(lens)}")
