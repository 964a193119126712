"""Parity of every HIP kernel against the CPU oracle, through the C ABI (needs an MI355X).

Tolerances: the kernels accumulate in fp32 and round once to bf16 where the oracle's
emulate_bf16 mode rounds, so results may differ by accumulation order only:  <= 1 bf16 ulp of
the output magnitude (2^-8 relative) plus a small absolute floor.
"""
import math

import numpy as np
import pytest
import torch

from oracle import model as om

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    e = Engine(DotsConfig.tiny(), max_batch=4, max_seq_len=512, max_patches=4096, max_prefill_tokens=2048)
    yield e
    e.close()


def bf(x):
    return x.to(torch.bfloat16)


def dev(x):
    return x.cuda().contiguous()


def close(got, ref, rel=2 ** -7, abs_=1e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = rel * ref.abs() + abs_ * max(1.0, float(ref.abs().max()))
    bad = err > tol
    assert not bad.any(), f"max err {err.max():.5f} (ref max {ref.abs().max():.4f}), {int(bad.sum())} / {bad.numel()} out of tolerance"


def run(eng, fn, *a):
    torch.cuda.synchronize()
    fn(*a)
    eng.synchronize()


@pytest.mark.parametrize("rows,dim", [(1, 256), (37, 768), (513, 1536), (5, 4096), (9, 512), (66, 1024)])
def test_rmsnorm(eng, rows, dim):
    g = torch.Generator().manual_seed(rows)
    x = bf(torch.randn(rows, dim, generator=g) * 3)
    w = bf(1 + 0.1 * torch.randn(dim, generator=g))
    xd, wd, yd = dev(x), dev(w), torch.empty(rows, dim, dtype=torch.bfloat16, device="cuda")
    run(eng, eng.op_rmsnorm, xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), rows, dim, 1e-5)
    close(yd, om.rms_norm(x.float(), w.float(), 1e-5, True), rel=2 ** -7)


@pytest.mark.parametrize("rows,dim", [(3, 256), (130, 1536)])
def test_layernorm(eng, rows, dim):
    g = torch.Generator().manual_seed(rows)
    x = bf(torch.randn(rows, dim, generator=g) * 2 + 0.5)
    w = bf(1 + 0.1 * torch.randn(dim, generator=g))
    b = bf(0.1 * torch.randn(dim, generator=g))
    xd, wd, bd, yd = dev(x), dev(w), dev(b), torch.empty(rows, dim, dtype=torch.bfloat16, device="cuda")
    run(eng, eng.op_layernorm, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), rows, dim, 1e-6)
    close(yd, om.layer_norm(x.float(), w.float(), b.float(), 1e-6, True))


def _pack_w13(gate, up):
    I, K = gate.shape
    return torch.stack([gate.view(I // 32, 32, K), up.view(I // 32, 32, K)], dim=1).reshape(2 * I, K)


@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (300, 256, 128), (1000, 384, 640), (129, 1536, 1536), (4097, 128, 4224),
                                   (257, 512, 64), (513, 768, 192), (255, 256, 320)])      # 2 / 6 / 10 K sub-tiles of the 256-wide kernel
@pytest.mark.parametrize("epi", [0, 1, 3, 4])
def test_gemm(eng, M, N, K, epi):
    from dots_ocr_amd import engine as E
    g = torch.Generator().manual_seed(M * 7 + N + epi)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = bf(torch.randn(N, generator=g))
    R = bf(torch.randn(M, N, generator=g))
    ref = A.float() @ W.float().t() + bias.float()
    if epi == E.EPI_RESIDUAL:
        ref = ref + R.float()
    if epi == E.EPI_GELU:
        ref = torch.nn.functional.gelu(ref)
    Ad, Wd, bd, Rd = dev(A), dev(W), dev(bias), dev(R)
    out = torch.empty(M, N, dtype=torch.float32 if epi == E.EPI_F32 else torch.bfloat16, device="cuda")
    run(eng, eng.op_gemm, Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), Rd.data_ptr() if epi == E.EPI_RESIDUAL else 0,
        out.data_ptr(), M, N, K, epi)
    if epi == E.EPI_F32:
        close(out, ref, rel=1e-4, abs_=1e-4)
    else:
        close(out, bf(ref))
    # transpose detection: the reference is not symmetric
    if M == N:
        assert (out.float().cpu() - ref.t()).abs().max() > 0.1


def test_gemm_vit_shape_is_repeatable_and_exact_everywhere(eng):
    """The ping-pong schedule orders LDS-DMA writes and ds_reads by counted vmcnt + barriers only: a race would show up as rare
    wrong tiles that come and go.  A ViT-shaped GEMM (48 K sub-tiles, 180 workgroups) run 6 times must be bit-identical every
    time and equal to the fp32 reference at EVERY element."""
    g = torch.Generator().manual_seed(99)
    M, N, K = 5000, 2304, 1536
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    ref = bf(A.float() @ W.float().t())
    Ad, Wd = dev(A), dev(W)
    outs = []
    for _ in range(6):
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        run(eng, eng.op_gemm, Ad.data_ptr(), Wd.data_ptr(), 0, 0, out.data_ptr(), M, N, K, 0)
        outs.append(out.cpu())
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16)), "GEMM result changes between runs"
    close(outs[0], ref)


def test_gemm_residual_in_place_and_no_bias(eng):
    from dots_ocr_amd import engine as E
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 256, 192
    A, W, R = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1), bf(torch.randn(M, N, generator=g))
    Ad, Wd, Rd = dev(A), dev(W), dev(R)
    run(eng, eng.op_gemm, Ad.data_ptr(), Wd.data_ptr(), 0, Rd.data_ptr(), Rd.data_ptr(), M, N, K, E.EPI_RESIDUAL)
    close(Rd, bf(R.float() + A.float() @ W.float().t()))


@pytest.mark.parametrize("M,I,K", [(70, 64, 128), (515, 512, 256)])
def test_gemm_swiglu(eng, M, I, K):
    from dots_ocr_amd import engine as E
    g = torch.Generator().manual_seed(M)
    A = bf(torch.randn(M, K, generator=g))
    gate, up = bf(torch.randn(I, K, generator=g) / math.sqrt(K)), bf(torch.randn(I, K, generator=g) / math.sqrt(K))
    W13 = _pack_w13(gate, up)
    ref = torch.nn.functional.silu(A.float() @ gate.float().t()) * (A.float() @ up.float().t())
    Ad, Wd = dev(A), dev(W13)
    out = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
    run(eng, eng.op_gemm, Ad.data_ptr(), Wd.data_ptr(), 0, 0, out.data_ptr(), M, 2 * I, K, E.EPI_SWIGLU)
    close(out, bf(ref))


def test_gemm_swiglu_epilogue_is_within_one_bf16_ulp_of_the_exact_quotient(eng):
    """the shared SwiGLU epilogue computes silu as rcp + one Newton step (gemm.hip: silu; -DGEMM_EXACT_SILU restores the IEEE
    division).  With identity weight blocks the GEMM hands the epilogue EXACT gate / up values, so the epilogue alone is compared with the
    exact x / (1 + e^-x) * u (fp64) over the whole range: at most one bf16 ulp apart, including x < -88 where e^-x overflows fp32 (the
    fast form returns a tiny negative or -0, never NaN / inf)."""
    from dots_ocr_amd import engine as E
    I = 128
    g = torch.Generator().manual_seed(77)
    x = torch.cat([torch.linspace(-120, 30, 1024), torch.randn(1024, generator=g) * 4, torch.tensor([-100., -88.5, -87., -20., -1e-3, 0., 1e-3, 20., 60.])])
    M = (x.numel() + I - 1) // I
    x = bf(torch.cat([x, torch.zeros(M * I - x.numel())]).view(M, I))
    u = bf(torch.randn(M, I, generator=g) * 2)
    A = torch.cat([x, u], 1)                                         # [M, 2 I]
    eye, zero = torch.eye(I), torch.zeros(I, I)
    gate, up = bf(torch.cat([eye, zero], 1)), bf(torch.cat([zero, eye], 1))
    out = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
    Ad, Wd = dev(A), dev(_pack_w13(gate, up))                        # (named: a temporary's block may be handed to the next allocation)
    run(eng, eng.op_gemm, Ad.data_ptr(), Wd.data_ptr(), 0, 0, out.data_ptr(), M, 2 * I, 2 * I, E.EPI_SWIGLU)
    got = out.float().cpu().double()
    xd, ud = x.double(), u.double()
    exact = xd / (1.0 + torch.exp(-xd)) * ud
    ref = exact.float().to(torch.bfloat16).double()
    assert torch.isfinite(got).all()
    ulp = torch.maximum(ref.abs(), torch.full_like(ref, 2.0 ** -126)) * 2.0 ** -7          # one bf16 ulp at the reference's magnitude (upper bound)
    bad = (got - ref).abs() > ulp + 1e-30                            # (+ the clamped exponent's tiny negatives below x = -80: ~1e-32, where the exact quotient underflows to -0)
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} SwiGLU outputs differ from the exact quotient by more than one bf16 ulp; worst x = {float(xd[bad][0])}"
    assert (got[xd < -88].abs() < 1e-30).all()


def _vt_reference(v, lens):
    """v [T, H, 128] -> V^T [H, 128, Tpad] with the kernel's padding and 16-key group order."""
    H = v.shape[1]
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    outs, t0 = [], 0
    for n in lens:
        npad = (n + 63) // 64 * 64
        vp = torch.zeros(npad, H, 128)
        vp[:n] = v[t0:t0 + n]
        idx = (torch.arange(npad) // 16 * 16).view(-1, 16)[:, 0:1] + perm.view(1, 16)       # position p holds key perm[p]
        outs.append(vp[idx.reshape(-1)])
        t0 += n
    return torch.cat(outs, 0).permute(1, 2, 0).contiguous()


def _attn_case(eng, lens, Hq, Hkv, causal, rope2d, seed):
    g = torch.Generator().manual_seed(seed)
    T = sum(lens)
    NQ = (Hq + 2 * Hkv) * 128
    qkv = bf(torch.randn(T, NQ, generator=g))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    Tpad = sum((n + 63) // 64 * 64 for n in lens)
    if rope2d:
        pos = torch.stack([torch.randint(0, 40, (T,), generator=g), torch.randint(0, 40, (T,), generator=g)], -1)
        inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float) / 64))
        fr = (pos.float().unsqueeze(-1) * inv).flatten(1)
        theta = 10000.0
    else:
        pos = torch.cat([torch.arange(n) for n in lens])
        inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))
        fr = pos.float().unsqueeze(-1) * inv
        theta = 1e6
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)
    x = qkv.float()
    q = x[:, :Hq * 128].view(T, Hq, 128)
    k = x[:, Hq * 128:(Hq + Hkv) * 128].view(T, Hkv, 128)
    v = x[:, (Hq + Hkv) * 128:].view(T, Hkv, 128)
    q_ref = bf(q * cos + om.rotate_half(q) * sin).float()
    k_ref = bf(k * cos + om.rotate_half(k) * sin).float()

    qd = torch.zeros(Hq, T, 128, dtype=torch.bfloat16, device="cuda")
    kd = torch.zeros(Hkv, T + 64, 128, dtype=torch.bfloat16, device="cuda")[:, :T]      # spare rows after the end
    kd = torch.zeros(Hkv * (T + 64) * 128, dtype=torch.bfloat16, device="cuda")
    vtd = torch.full((Hkv, 128, Tpad), 7.0, dtype=torch.bfloat16, device="cuda")
    qkvd = dev(qkv)
    run(eng, eng.op_qkv_rope_split, qkvd.data_ptr(), qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), cu,
        pos.numpy().astype(np.int32), Hq, Hkv, rope2d, theta)
    k_got = kd[: Hkv * T * 128].view(Hkv, T, 128)
    close(qd.permute(1, 0, 2), q_ref, rel=2 ** -7, abs_=2e-3)
    close(k_got.permute(1, 0, 2), k_ref, rel=2 ** -7, abs_=2e-3)
    assert torch.equal(vtd.float().cpu(), _vt_reference(v, lens))

    out = torch.zeros(T, Hq * 128, dtype=torch.bfloat16, device="cuda")
    scale = 1 / math.sqrt(128)
    run(eng, eng.op_flash_attn, qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), cu, Hq, Hkv, causal, scale)
    ref = torch.empty(T, Hq, 128)
    t0 = 0
    rep = Hq // Hkv
    for n in lens:
        ref[t0:t0 + n] = om._attention(q_ref[t0:t0 + n].transpose(0, 1),
                                       k_ref[t0:t0 + n].transpose(0, 1).repeat_interleave(rep, 0),
                                       v[t0:t0 + n].transpose(0, 1).repeat_interleave(rep, 0), scale, causal, True).transpose(0, 1)
        t0 += n
    close(out.view(T, Hq, 128), ref, rel=2 ** -6, abs_=4e-3)


@pytest.mark.parametrize("lens", [[64], [200, 64, 1, 333], [128, 129, 127], [1000]])
def test_vision_rope_split_and_flash_attn(eng, lens):
    _attn_case(eng, lens, 2, 2, False, True, seed=sum(lens))


@pytest.mark.parametrize("lens", [[130, 77], [64, 65, 1], [700]])
def test_lm_rope_split_and_causal_gqa_flash_attn(eng, lens):
    _attn_case(eng, lens, 6, 1, True, False, seed=sum(lens) + 1)


@pytest.mark.parametrize("lens,Hq,Hkv,K,rope2d,with_bias", [
    ([300, 212], 12, 12, 1536, True, False),        # the tower's shape: N = 4608, ragged last m-tile (512 = 2 x 256 exactly: full tiles only)
    ([700, 77, 1], 12, 12, 256, True, False),       # 778 rows: a ragged last m-tile, a one-token sequence
    ([130, 515], 12, 2, 1536, False, True),         # the LM prefill's shape: N = 2048, qkv bias, 1-D rope
    ([64], 2, 2, 192, True, True),                  # fewer rows than a wave tile
    ([4099, 4093], 12, 12, 192, True, False),       # 25 M q / k elements: the rare roundings where a different fma contraction would show
    ([8191], 12, 2, 192, False, True),
])
def test_qkv_gemm_with_the_rope_epilogue_equals_gemm_then_split_bitwise(eng, lens, Hq, Hkv, K, rope2d, with_bias):
    """Round 6: the q / k heads leave the qkv GEMM rotated and head-major (gemm.hip: w4_epilogue_qkrope), the split kernel only transposes v.
    The fused pair must give the bits of GEMM -> qkv_rope_split_kernel (dots_op_qkv_proj_rope runs either), which the tests above hold to the oracle."""
    eng.set_gemm_plan(1)                              # the fused kernel belongs to the one-wave-per-SIMD plan (the process default)
    g = torch.Generator().manual_seed(sum(lens) + K)
    T, N = sum(lens), (Hq + 2 * Hkv) * 128
    x = dev(bf(torch.randn(T, K, generator=g)))
    w = dev(bf(torch.randn(N, K, generator=g) * (1.5 / math.sqrt(K))))
    b = dev(bf(torch.randn(N, generator=g))) if with_bias else None
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    Tpad = sum((n + 63) // 64 * 64 for n in lens)
    if rope2d:
        pos = torch.stack([torch.randint(0, 120, (T,), generator=g), torch.randint(0, 170, (T,), generator=g)], -1).numpy().astype(np.int32)
        theta = 10000.0
    else:
        pos = torch.cat([torch.arange(n) for n in lens]).numpy().astype(np.int32)
        theta = 1e6
    outs = []
    for fused in (0, 1):
        ws = torch.full((T, N), 3.0, dtype=torch.bfloat16, device="cuda")
        qd = torch.full((Hq, T, 128), 5.0, dtype=torch.bfloat16, device="cuda")
        kd = torch.full((Hkv * (T + 64) * 128,), 5.0, dtype=torch.bfloat16, device="cuda")
        vtd = torch.full((Hkv, 128, Tpad), 7.0, dtype=torch.bfloat16, device="cuda")
        run(eng, eng.op_qkv_proj_rope, x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else 0, ws.data_ptr(), qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(),
            cu, pos, K, Hq, Hkv, rope2d, theta, fused)
        outs.append((qd.view(torch.int16).cpu(), kd.view(torch.int16).cpu(), vtd.view(torch.int16).cpu(), ws[:, (Hq + Hkv) * 128:].contiguous().view(torch.int16).cpu()))
    for name, a_, b_ in zip(("q", "k", "v^T", "v columns of the workspace"), outs[0], outs[1]):
        assert torch.equal(a_, b_), f"{name}: the fused path differs from GEMM + split ({int((a_ != b_).sum())} of {a_.numel()} elements)"
    assert not torch.equal(outs[0][0], torch.full_like(outs[0][0], outs[0][0][0, 0, 0].item())), "q was not written"


def test_flash_attn_online_softmax_rescale_branch(eng):
    """A key far above the rest arriving in a LATE tile forces the running-max rescale (guide §5.4 rule 26)."""
    g = torch.Generator().manual_seed(11)
    n, H = 300, 2
    q = bf(torch.randn(n, H, 128, generator=g))
    k = bf(torch.randn(n, H, 128, generator=g))
    v = bf(torch.randn(n, H, 128, generator=g))
    k[250] = bf(q[7] * 4)                  # spike for query 7 in the 4th tile
    cu = np.array([0, n], np.int32)
    qd = dev(q.permute(1, 0, 2))
    kd = torch.zeros(H * (n + 64) * 128, dtype=torch.bfloat16, device="cuda")
    kd[: H * n * 128] = dev(k.permute(1, 0, 2)).reshape(-1)
    vtd = dev(bf(_vt_reference(v.float(), [n])))
    out = torch.zeros(n, H * 128, dtype=torch.bfloat16, device="cuda")
    scale = 1 / math.sqrt(128)
    run(eng, eng.op_flash_attn, qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), cu, H, H, False, scale)
    ref = om._attention(q.float().transpose(0, 1), k.float().transpose(0, 1), v.float().transpose(0, 1), scale, False, True)
    close(out.view(n, H, 128), ref.transpose(0, 1), rel=2 ** -6, abs_=4e-3)


@pytest.mark.parametrize("w,h", [(1654, 2339), (333, 517), (100, 40), (1344, 1344), (3000, 4200), (28, 28)])
def test_gpu_preprocess_is_bit_identical_to_host_pillow_path(eng, w, h):
    """uint8 page -> float32 patches on the GPU (Pillow-exact fixed-point bicubic + normalise + patchify)
    == dots_ocr_amd.image_utils.preprocess_image, bit for bit (upscale, downscale past max_pixels, identity)."""
    from PIL import Image
    from dots_ocr_amd.image_utils import preprocess_image
    rng = np.random.default_rng(w + h)
    arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref, thw = preprocess_image(Image.fromarray(arr, "RGB"))
    out = torch.empty(ref.shape, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    got_thw = eng.preprocess_image(arr, out.data_ptr())
    assert got_thw == thw
    got = out.cpu().numpy()
    bad = got.view(np.uint32) != ref.view(np.uint32)
    assert not bad.any(), f"{int(bad.sum())} / {bad.size} values differ, max abs diff {np.abs(got - ref).max():.3e}"
