"""Parity at BASELINE.json's FULL sizes (A4@200dpi: 19 824 patches, 12 heads; T ~ 5.2 k LM tokens), where the
CPU oracle cannot run the whole tensor in seconds.  Uses size-independent properties of the operators plus
exact oracle checks on SAMPLED rows (one query row against all 19 824 keys is cheap on the CPU).
"""
import math

import numpy as np
import pytest
import torch

from oracle import model as om

pytestmark = pytest.mark.gpu

N_A4 = 19824


@pytest.fixture(scope="module")
def eng():
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    e = Engine(DotsConfig.tiny(), max_batch=2, max_seq_len=256, max_patches=256, max_prefill_tokens=256)
    yield e
    e.close()


def _vt(v):
    """v [n, H, 128] (cuda, bf16) -> kernel V^T layout [H, 128, npad] (keys permuted inside 16-groups, zero pad)."""
    n, H, _ = v.shape
    npad = (n + 63) // 64 * 64
    vp = torch.zeros(npad, H, 128, dtype=v.dtype, device=v.device)
    vp[:n] = v
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15], device=v.device)
    idx = (torch.arange(npad, device=v.device) // 16 * 16).view(-1, 16)[:, :1] + perm.view(1, 16)
    return vp[idx.reshape(-1)].permute(1, 2, 0).contiguous()


def _flash(eng, q, k, v, causal=False, Hkv=None):
    """q [n,H,128], k/v [n,Hkv,128] cuda bf16 -> out [n, H, 128]."""
    n, H, _ = q.shape
    Hkv = Hkv or H
    qd = q.permute(1, 0, 2).contiguous()
    kd = torch.zeros(Hkv, n + 64, 128, dtype=torch.bfloat16, device="cuda")
    kd[:, :n] = k.permute(1, 0, 2)
    kd = kd[:, :n].contiguous()
    vt = _vt(v)
    out = torch.zeros(n, H * 128, dtype=torch.bfloat16, device="cuda")
    cu = np.array([0, n], np.int32)
    torch.cuda.synchronize()
    eng.op_flash_attn(qd.data_ptr(), kd.data_ptr(), vt.data_ptr(), out.data_ptr(), cu, H, Hkv, causal, 1 / math.sqrt(128))
    eng.synchronize()
    return out.view(n, H, 128)


def test_flash_attn_a4_sampled_rows_match_oracle(eng):
    g = torch.Generator(device="cuda").manual_seed(1)
    H = 12
    q = torch.randn(N_A4, H, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(N_A4, H, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(N_A4, H, 128, device="cuda", generator=g).bfloat16()
    out = _flash(eng, q, k, v).float().cpu()
    rows = [0, 1, 31, 32, 127, 128, 4097, 9999, N_A4 - 49, N_A4 - 1]          # block edges, ragged last tile
    qs, kf, vf = q[rows].float().cpu(), k.float().cpu(), v.float().cpu()
    ref = om._attention(qs.transpose(0, 1), kf.transpose(0, 1), vf.transpose(0, 1), 1 / math.sqrt(128), False, True).transpose(0, 1)
    err = (out[rows] - ref).abs().max().item()
    assert err < 4e-3 + 2 ** -6 * ref.abs().max().item(), err


def _flash_packed(eng, qs, ks, vs):
    """A packed batch of sequences (lists of [n_i, H, 128] cuda bf16), laid out as the engine lays the ViT batch out: q / k head-major over
    the packed tokens, V^T per sequence padded to a multiple of 64 keys.  -> list of [n_i, H, 128]."""
    H = qs[0].shape[1]
    lens = [q.shape[0] for q in qs]
    T = sum(lens)
    qd = torch.cat(qs).permute(1, 0, 2).contiguous()
    kd = torch.zeros(H * T + 64, 128, dtype=torch.bfloat16, device="cuda")          # head stride T (the engine's layout) + the 64 spare rows
    kd[:H * T] = torch.cat(ks).permute(1, 0, 2).reshape(H * T, 128)
    vt = torch.cat([_vt(v) for v in vs], dim=2).contiguous()
    out = torch.zeros(T, H * 128, dtype=torch.bfloat16, device="cuda")
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    torch.cuda.synchronize()
    eng.op_flash_attn(qd.data_ptr(), kd.data_ptr(), vt.data_ptr(), out.data_ptr(), cu, H, H, False, 1 / math.sqrt(128))
    eng.synchronize()
    return list(out.view(T, H, 128).split(lens))


def _check_rows(out, q, k, v, rows):
    qs, kf, vf = q[rows].float().cpu(), k.float().cpu(), v.float().cpu()
    ref = om._attention(qs.transpose(0, 1), kf.transpose(0, 1), vf.transpose(0, 1), 1 / math.sqrt(128), False, True).transpose(0, 1)
    err = (out[rows].float().cpu() - ref).abs().max().item()
    assert err < 4e-3 + 2 ** -6 * ref.abs().max().item(), (err, q.shape[0])


@pytest.mark.parametrize("n", [39648, 56644])           # an A3 page of the mixed64 set (236 x 168 patches); the largest page the reference admits (238 x 238)
def test_flash_attn64_sampled_rows_match_oracle_above_a4(eng, n):
    """VERDICT r4 #4a: flash_attn64_kernel against the oracle at the sequence lengths above A4 that the workloads contain — 620 / 886 KV tiles
    per query block, a ragged last tile (39 648 = 619 x 64 + 32, 56 644 = 885 x 64 + 4), 4 heads (the kernel's work items are per head)."""
    g = torch.Generator(device="cuda").manual_seed(n)
    H = 4
    q = torch.randn(n, H, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(n, H, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(n, H, 128, device="cuda", generator=g).bfloat16()
    out = _flash(eng, q, k, v)
    _check_rows(out, q, k, v, [0, 63, 64, 255, 256, 20001, n // 2, n - 65, n - 33, n - 1])


def test_flash_attn64_ragged_packed_batch_of_the_mixed64_sizes(eng):
    """A packed batch with one page of every size class of BASELINE configs[3] (bench.py --workload mixed64): A3, A4, 1700x2250, 1344x1344,
    946x1024, 583x550 -> 39 648 / 19 824 / 19 520 / 9 216 / 5 032 / 1 680 patches in ONE launch; sampled rows of every sequence (first and
    last query block, the rows next to the sequence boundaries) against the oracle run on that sequence alone."""
    from dots_ocr_amd.image_utils import smart_resize
    sizes = [(2339, 3308), (1654, 2339), (1700, 2250), (1344, 1344), (946, 1024), (583, 550)]
    lens = []
    for w, h in sizes:
        rh, rw = smart_resize(h, w)
        lens.append((rh // 14) * (rw // 14))
    assert lens[:4] == [39648, 19824, 19520, 9216] and lens[5] == 1680, lens
    g = torch.Generator(device="cuda").manual_seed(64)
    H = 2
    qs = [torch.randn(n, H, 128, device="cuda", generator=g).bfloat16() for n in lens]
    ks = [torch.randn(n, H, 128, device="cuda", generator=g).bfloat16() for n in lens]
    vs = [torch.randn(n, H, 128, device="cuda", generator=g).bfloat16() for n in lens]
    outs = _flash_packed(eng, qs, ks, vs)
    for out, q, k, v in zip(outs, qs, ks, vs):
        n = q.shape[0]
        _check_rows(out, q, k, v, sorted({0, 1, 63, 64, min(255, n - 1), min(256, n - 1), n // 2, n - 64, n - 2, n - 1}))


def test_flash_attn_a4_properties(eng):
    g = torch.Generator(device="cuda").manual_seed(2)
    H = 12
    q = torch.randn(N_A4, H, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(N_A4, H, 128, device="cuda", generator=g).bfloat16()
    # (1) softmax weights sum to one: constant V rows come back unchanged (up to bf16 rounding of P and O)
    c = torch.randn(1, H, 128, device="cuda", generator=g).bfloat16()
    out = _flash(eng, q, k, c.expand(N_A4, H, 128).contiguous())
    assert (out.float() - c.float()).abs().max().item() < 0.03
    # (2) bidirectional attention is invariant to a permutation of the keys (K and V rows together)
    v = torch.randn(N_A4, H, 128, device="cuda", generator=g).bfloat16()
    perm = torch.randperm(N_A4, device="cuda", generator=g)
    a = _flash(eng, q, k, v).float()
    b = _flash(eng, q, k[perm].contiguous(), v[perm].contiguous()).float()
    assert (a - b).abs().max().item() < 0.02            # different tile order -> different rounding, same value
    # (3) linear in V
    v2 = torch.randn(N_A4, H, 128, device="cuda", generator=g).bfloat16()
    s = _flash(eng, q, k, (v.float() + v2.float()).bfloat16()).float()
    b2 = _flash(eng, q, k, v2).float()
    assert (s - (a + b2)).abs().max().item() < 0.05
    # (4) one dominant key: row i attends (almost) only to key j
    kk = k.clone()
    kk[777] = (q[5] * 8).bfloat16()
    o = _flash(eng, q, kk, v).float()
    assert (o[5] - v[777].float()).abs().max().item() < 0.02


def test_causal_gqa_full_prompt_length_sampled_rows(eng):
    g = torch.Generator(device="cuda").manual_seed(3)
    T, Hq, Hkv = 5200, 12, 2
    q = torch.randn(T, Hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, Hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, Hkv, 128, device="cuda", generator=g).bfloat16()
    out = _flash(eng, q, k, v, causal=True, Hkv=Hkv).float().cpu()
    # first row sees only key 0
    assert (out[0] - v[0].float().cpu().repeat_interleave(6, 0)).abs().max().item() < 0.02
    kf, vf = k.float().cpu().repeat_interleave(6, 1), v.float().cpu().repeat_interleave(6, 1)
    for r in (1, 63, 64, 127, 128, 191, 192, 255, 256, 257, 2600, 4863, 4864, 5119, 5120, T - 1):      # every wave boundary of a 256-row block, first and last blocks
        ref = om._attention(q[r:r + 1].float().cpu().transpose(0, 1), kf[:r + 1].transpose(0, 1), vf[:r + 1].transpose(0, 1),
                            1 / math.sqrt(128), False, True).transpose(0, 1)[0]
        assert (out[r] - ref).abs().max().item() < 4e-3 + 2 ** -6 * ref.abs().max().item(), r


def test_gemm_a4_rows_sampled_and_linearity(eng):
    """The ViT qkv GEMM shape at one A4 page (M = 19 824, N = 4 608, K = 1 536): sampled rows vs fp32, and
    linearity C(A1 + A2) = C(A1) + C(A2) over the whole output (a checksum-of-checksums style property)."""
    g = torch.Generator(device="cuda").manual_seed(4)
    M, N, K = N_A4, 4608, 1536
    A1 = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    C1 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    eng.op_gemm(A1.data_ptr(), W.data_ptr(), 0, 0, C1.data_ptr(), M, N, K, 4)          # fp32 output
    eng.synchronize()
    rows = [0, 127, 128, 255, 256, 10007, M - 1]
    ref = A1[rows].float() @ W.float().t()
    assert (C1[rows] - ref).abs().max().item() < 2e-3
    A2 = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    A3 = (A1.float() + A2.float())                       # exact in fp32; representable in bf16 only approximately,
    A3b = A3.bfloat16()                                  # so compare against the bf16-rounded sum's own product
    C2 = torch.empty_like(C1); C3 = torch.empty_like(C1)
    torch.cuda.synchronize()
    eng.op_gemm(A2.data_ptr(), W.data_ptr(), 0, 0, C2.data_ptr(), M, N, K, 4)
    eng.op_gemm(A3b.data_ptr(), W.data_ptr(), 0, 0, C3.data_ptr(), M, N, K, 4)
    eng.synchronize()
    resid = (A3b.float() - A3)                           # rounding of the operand sum, propagated exactly
    corr = resid[rows] @ W.float().t()
    assert (C3[rows] - (C1[rows] + C2[rows] + corr)).abs().max().item() < 3e-3
    # column checksum over ALL rows: sum_m C[m][n] == (sum_m A[m]) . W[n]
    col = C1.double().sum(0)
    ref_col = (A1.double().sum(0, keepdim=True) @ W.double().t())[0]
    assert (col - ref_col).abs().max().item() < 2e-2 * max(1.0, ref_col.abs().max().item())


def test_full_size_model_a4_pages_deterministic_and_batch_invariant():
    """The real architecture (42-layer 1536-wide ViT, 28-layer LM, vocab 151 936; seeded random weights) on two
    synthetic A4@200dpi pages: same tokens run-to-run, and a page decodes identically alone or inside a batch —
    the property page-level data parallelism relies on."""
    import os
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.image_utils import preprocess_image
    from dots_ocr_amd.synthetic import A4_200DPI, synth_page, synth_prompt_ids
    from dots_ocr_amd.weights import random_state_dict
    cfg = DotsConfig()
    from shared_weights import full_sd
    sd = full_sd(0)
    eng = Engine(cfg, max_batch=2, max_seq_len=14600, max_patches=57600 + 64, max_prefill_tokens=14600)
    eng.load_state_dict(sd)
    del sd
    feats, grids = zip(*(preprocess_image(synth_page(i, A4_200DPI)) for i in range(2)))
    pv, grid = np.concatenate(feats, 0), np.asarray(grids, np.int64)
    assert pv.shape == (2 * N_A4, 588)
    prompts = [synth_prompt_ids(cfg, 4956, seed=i) for i in range(2)]
    ids, lens = np.concatenate(prompts), np.array([5200, 5200], np.int32)
    a, la = eng.generate(ids, lens, pv, grid, max_new_tokens=8)
    b, _ = eng.generate(ids, lens, pv, grid, max_new_tokens=8)
    assert la.tolist() == [8, 8] and np.array_equal(a, b)
    c, _ = eng.generate(prompts[1], lens[1:], pv[N_A4:], grid[1:], max_new_tokens=8)
    assert np.array_equal(c[0], a[1])
    st = eng.stats()
    assert st["vit_patches"] == N_A4 and abs(st["vit_flops"] / 150.0e12 - 1) < 0.01      # 150.0 TFLOP per A4 page

    # ragged batch at real dimensions (BASELINE configs 3-4 geometry): 1344x1344 "high-res" page (9216 patches) and a
    # 583x550 chart-sized page (-> 588x560, 1680 patches); each page must decode as it does alone
    from dots_ocr_amd.synthetic import HIGH_RES
    sizes = [HIGH_RES, (583, 550)]
    feats2, grids2 = zip(*(preprocess_image(synth_page(10 + i, sz)) for i, sz in enumerate(sizes)))
    assert [f.shape[0] for f in feats2] == [9216, 1680]          # 2304 and 420 vision tokens (SURVEY §4 fixtures)
    pv2, grid2 = np.concatenate(feats2, 0), np.asarray(grids2, np.int64)
    prompts2 = [synth_prompt_ids(cfg, f.shape[0] // 4, n_text_tokens=37 + 11 * i, seed=20 + i) for i, f in enumerate(feats2)]
    lens2 = np.array([len(p) for p in prompts2], np.int32)
    m, _ = eng.generate(np.concatenate(prompts2), lens2, pv2, grid2, max_new_tokens=6)
    off = 0
    for i, f in enumerate(feats2):
        single, _ = eng.generate(prompts2[i], lens2[i:i + 1], f, grid2[i:i + 1], max_new_tokens=6)
        assert np.array_equal(single[0], m[i]), i
        off += f.shape[0]

    # the largest page the reference admits: 4500x4500 px is above MAX_PIXELS and smart_resize brings it to 3332x3332
    # (SURVEY §8(c) known answer) -> 238 x 238 = 56 644 patches, 14 161 vision tokens, one 56 644-key attention per head
    from dots_ocr_amd.image_utils import smart_resize
    assert smart_resize(4500, 4500) == (3332, 3332)
    big, gbig = preprocess_image(synth_page(30, (4500, 4500)))
    assert big.shape == (56644, 588) and list(gbig) == [1, 238, 238]
    pbig = synth_prompt_ids(cfg, 14161, n_text_tokens=40, seed=31)
    lbig = np.array([len(pbig)], np.int32)
    x, lx = eng.generate(pbig, lbig, big, np.asarray([gbig], np.int64), max_new_tokens=4)
    y, _ = eng.generate(pbig, lbig, big, np.asarray([gbig], np.int64), max_new_tokens=4)
    assert lx.tolist() == [4] and np.array_equal(x, y) and (x >= 0).all() and (x < cfg.vocab_size).all()
    st = eng.stats()
    n = 56644
    flops = 42 * (2 * n * (4 * 1536 ** 2 + 3 * 1536 * 4224) + 4 * n * n * 1536) + 2 * n * 588 * 1536 + 2 * (n // 4) * (6144 ** 2 + 6144 * 1536)
    assert st["vit_patches"] == n and abs(st["vit_flops"] / flops - 1) < 0.01            # SURVEY §8(d) ViT formula (~965 TFLOP)
    from dots_ocr_amd.engine import DotsEngineError
    with pytest.raises(DotsEngineError, match="max_patches"):                             # more patches than the ViT workspace holds
        eng.vit_forward(np.zeros((244 * 238, 588), np.float32), np.asarray([[1, 244, 238]], np.int64))
    eng.close()
