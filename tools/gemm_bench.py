#!/usr/bin/env python3
"""Times the GEMM kernel through the C ABI on the ViT / prefill shapes of the A4 bench (8 pages: M = 158 592 patches, 41 600 prompt
tokens).  DOTS_OCR_LIB selects a variant build (tools/build_variant_gemm.sh), DOTS_OCR_GEMM_LOCKSTEP=1 the previous schedule.
Random operands (zero-filled ones clock higher: MI355X_MICROARCH.md, DVFS)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from dots_ocr_amd.config import DotsConfig  # noqa: E402
from dots_ocr_amd.engine import EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, Engine  # noqa: E402

SHAPES = [("vit qkv", 158592, 4608, 1536, EPI_NONE), ("vit proj", 158592, 1536, 1536, EPI_RESIDUAL), ("vit fc13", 158592, 8448, 1536, EPI_SWIGLU),
          ("vit fc2", 158592, 1536, 4224, EPI_RESIDUAL), ("lm gate|up", 41600, 17920, 1536, EPI_SWIGLU), ("lm down", 41600, 1536, 8960, EPI_RESIDUAL)]


def main():
    eng = Engine(DotsConfig.tiny(), max_batch=1, max_seq_len=256, max_patches=256, max_prefill_tokens=256)
    g = torch.Generator(device="cuda").manual_seed(0)
    tot_ms, tot_fl = 0.0, 0.0
    for name, M, N, K, epi in SHAPES:
        A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * 0.05).bfloat16()
        ldc = N // 2 if epi == EPI_SWIGLU else N
        C = torch.zeros(M, ldc, dtype=torch.bfloat16, device="cuda")
        R = C.data_ptr() if epi == EPI_RESIDUAL else None
        torch.cuda.synchronize()
        for _ in range(2):
            eng.op_gemm(A.data_ptr(), W.data_ptr(), None, R, C.data_ptr(), M, N, K, epi)
        eng.synchronize()
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.op_gemm(A.data_ptr(), W.data_ptr(), None, R, C.data_ptr(), M, N, K, epi)
        eng.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        fl = 2.0 * M * N * K
        w = 42 if name.startswith("vit") else 28
        tot_ms += ms * w
        tot_fl += fl * w
        print(f"{name:12s} M={M} N={N} K={K}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
        del A, W, C
    print(f"weighted (42 ViT blocks + 28 LM layers): {tot_ms:.1f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
    eng.close()


if __name__ == "__main__":
    main()
