"""gfx950 MFMA fragment maps + LDS-DMA, proven on hardware with asymmetric operands.
Every MFMA kernel in the engine assumes the maps documented in csrc/probe_mfma.hip."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16_bits(x):
    import torch
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("which,m,k", [(0, 32, 16), (1, 16, 32)])
def test_mfma_maps(which, m, k):
    import torch
    from dots_ocr_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(7 + which)
    A = torch.randn(m, k, generator=g).to(torch.bfloat16)
    Bt = torch.randn(m, k, generator=g).to(torch.bfloat16)       # Bt[j][k] = B[k][j]
    ref = A.float() @ Bt.float().t()                              # asymmetric: ref != ref.T
    Ad, Bd = A.cuda(), Bt.cuda()
    D = torch.zeros(m, m, dtype=torch.float32, device="cuda")
    rc = lib.dots_probe_mfma(which, ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()),
                             ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    err = (D.cpu() - ref).abs().max().item()
    assert err < 1e-3, f"MFMA map mismatch (which={which}): max err {err}"
    assert (D.cpu() - ref.t()).abs().max().item() > 1e-2      # the check is transpose-detecting


def test_mfma_fp8_32x32x64_pairs_equal_byte_positions():
    """v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit scales: lane l feeds bytes [32 (l >> 5), +32) of row l & 31 for
    A and for B -> D = A B^T.  The block-scaled datapath does not keep every bit of a 64-term sum (measured 2.5e-3 on sums of
    magnitude ~30, i.e. ~1e-4 relative, against ~1e-7 for fp32 accumulation): the tolerance allows that, a wrong pairing is O(10)."""
    import torch
    from dots_ocr_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(31)
    A = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
    Bt = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
    ref = A.float() @ Bt.float().t()
    Ad, Bd = A.cuda(), Bt.cuda()
    D = torch.zeros(32, 32, dtype=torch.float32, device="cuda")
    rc = lib.dots_probe_mfma(3, ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()),
                             ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    err = (D.cpu() - ref).abs().max().item()
    assert err < 2e-2, f"fp8 MFMA map mismatch: max err {err}"
    assert (D.cpu() - ref.t()).abs().max().item() > 1.0


def test_lds_dma():
    import torch
    from dots_ocr_amd import _lib
    lib = _lib.load()
    x = torch.arange(256, dtype=torch.int32).cuda() * 3 + 1
    y = torch.zeros(256, dtype=torch.int32, device="cuda")
    rc = lib.dots_probe_mfma(2, ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(y.data_ptr()),
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(x, y)
