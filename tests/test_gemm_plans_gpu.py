"""The two launch plans of the 256-wide bf16 GEMM (dots_set_gemm_plan): 0 = 8 waves, ping-pong halves (round 2);
1 = one wave per SIMD, 128 x 128 wave tiles, K tiles of 64 through a 5-unit LDS-DMA ring (round 5).

Both issue the same MFMAs in the same k order per output element, so their results must be BIT-identical; plan 1 is also
checked against the fp32 reference and for repeatability (its LDS ring is ordered by counted vmcnt + one barrier per K tile
only: a race would show up as rare wrong tiles).  K values cover every phase of the 5-unit ring (3 .. 8 K tiles and the
real 24 / 66 / 140), M values the clamped row tail, N the 1 / 2 / 9-tile cases.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    e = Engine(DotsConfig.tiny(), max_batch=2, max_seq_len=256, max_patches=256, max_prefill_tokens=256)
    yield e
    e.set_gemm_plan(1)                                # the process-wide default (round 5); a 0 here left every later module of the session on the round-2 kernel
    e.close()


def _run(eng, plan, A, W, bias, R, M, N, K, epi):
    from dots_ocr_amd import engine as E
    eng.set_gemm_plan(plan)
    ldc = N // 2 if epi == E.EPI_SWIGLU else N
    out = torch.zeros(M, ldc, dtype=torch.float32 if epi == E.EPI_F32 else torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    eng.op_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr() if bias is not None else 0, R.data_ptr() if epi == E.EPI_RESIDUAL else 0,
                out.data_ptr(), M, N, K, epi)
    eng.synchronize()
    return out


def _bits(t):
    return t.view(torch.int32 if t.dtype == torch.float32 else torch.int16)


@pytest.mark.parametrize("M,N,K", [(1, 256, 192), (255, 256, 256), (257, 512, 320), (513, 256, 384), (300, 768, 448), (1000, 256, 512),
                                   (129, 1536, 1536), (700, 512, 4224), (260, 256, 8960)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_one_wave_per_simd_plan_equals_ping_pong_plan_bitwise(eng, M, N, K, epi):
    g = torch.Generator().manual_seed(M * 131 + N + K + epi)
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().cuda()
    bias = torch.randn(N, generator=g).bfloat16().cuda()
    R = torch.randn(M, N, generator=g).bfloat16().cuda()
    o0 = _run(eng, 0, A, W, bias, R, M, N, K, epi)
    o1 = _run(eng, 1, A, W, bias, R, M, N, K, epi)
    assert torch.equal(_bits(o0), _bits(o1)), f"plans differ in {(o0 != o1).sum().item()} elements"
    if epi in (0, 4):                                       # and both are right (the parity tests proper run under plan 0)
        ref = A.float() @ W.float().t() + bias.float()
        got = o1.float()
        tol = 2 ** -7 * ref.abs() + 1e-3 * max(1.0, float(ref.abs().max()))
        assert ((got - ref).abs() <= tol).all()


def test_one_wave_per_simd_plan_is_repeatable_on_the_vit_shape(eng):
    g = torch.Generator().manual_seed(99)
    M, N, K = 5000, 2304, 1536
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().cuda()
    ref = _run(eng, 0, A, W, None, None, M, N, K, 0)
    for _ in range(6):
        out = _run(eng, 1, A, W, None, None, M, N, K, 0)
        assert torch.equal(_bits(out), _bits(ref)), "GEMM result changes between runs / plans"
