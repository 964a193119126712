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
