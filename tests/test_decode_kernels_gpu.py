"""Oracle parity of every kernel of the decode step AT THE BASELINE DIMENSIONS (H = 1536, I = 8960, V = 151 936, GQA 12:2,
head_dim 128), one kernel at a time through the C ABI (dots_op_dec_* / dots_op_decode_attn), against oracle/model.py's
restatement of transformers Qwen2 (reference call site dots_ocr/parser.py:110).

Covered: batch sizes {1, 8, 9, 16} (8-row and 16-row X images), context lengths {1, 63, 64, 65, 5 200, 6 223, 32 767}
(first key of a page, last key of a page, ragged last page, the bench's contexts, the default 32 k capacity), KV splits
{1, 3, 25, 64} with empty splits, page tables that are NOT the identity, K/V append at a page boundary, untouched neighbours.

Tolerances: the kernels accumulate in fp32 and round once to bf16 where the oracle's emulate_bf16 mode rounds, so they may
differ from it by accumulation order only: <= 1 bf16 ulp of the output magnitude (2^-8 relative) plus a small absolute
floor; fp32 logits 1e-3 x max|logit| absolute (a 1-ulp rsqrt difference flips a few bf16 roundings of the normalised row).
"""
import math

import numpy as np
import pytest
import torch

from oracle import model as om

pytestmark = pytest.mark.gpu

H, I, V, HQ, HKV = 1536, 8960, 151936, 12, 2
EPS, THETA = 1e-6, 1e6


@pytest.fixture(scope="module")
def eng():
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    e = Engine(DotsConfig.tiny(), max_batch=2, max_seq_len=256, max_patches=256, max_prefill_tokens=256)
    yield e
    e.close()


def bf(x):
    return x.to(torch.bfloat16)


def dev(x):
    return x.cuda().contiguous()


def close(got, ref, rel=2 ** -7, abs_=1e-3, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    tol = rel * ref.abs() + abs_ * max(1.0, float(ref.abs().max()))
    bad = err > tol
    assert not bad.any(), f"{what}: max err {err.max():.5f} (ref max {ref.abs().max():.4f}), {int(bad.sum())} / {bad.numel()} out of tolerance"


# ---- KV page layout (csrc/decode.hip header), restated for the tests
def _page_index_maps():
    key = torch.arange(64).view(64, 1)
    d = torch.arange(128).view(1, 128)
    k_idx = (((key >> 4) * 4 + (d >> 5)) * 64 + ((d >> 3) & 3) * 16 + (key & 15)) * 8 + (d & 7)
    kk = key & 31
    v_idx = (((key >> 5) * 8 + (d >> 4)) * 64 + ((kk >> 2) & 3) * 16 + (d & 15)) * 8 + 4 * (kk >> 4) + (kk & 3)
    assert sorted(k_idx.flatten().tolist()) == list(range(8192)) and sorted(v_idx.flatten().tolist()) == list(range(8192))
    return k_idx, v_idx


K_IDX, V_IDX = _page_index_maps()


def pack_pages(pool, table_row, K, Vv):
    """K, V [n, Hkv, 128] bf16 of ONE sequence -> its pages of pool [pages, Hkv, 2, 8192] (cpu bf16) through table_row."""
    n = K.shape[0]
    for p in range((n + 63) // 64):
        m = min(64, n - p * 64)
        pg = int(table_row[p])
        for hk in range(K.shape[1]):
            flatk = pool[pg, hk, 0]
            flatv = pool[pg, hk, 1]
            flatk[K_IDX[:m].reshape(-1)] = K[p * 64:p * 64 + m, hk].reshape(-1)
            flatv[V_IDX[:m].reshape(-1)] = Vv[p * 64:p * 64 + m, hk].reshape(-1)


def _rope(x, pos):
    """x [B, heads, 128] fp32, pos [B] -> rotated (transformers Qwen2 rotate_half convention), fp32."""
    cos, sin = om.lm_rope_cos_sin(pos, 128, THETA)
    return x * cos.unsqueeze(1) + om.rotate_half(x) * sin.unsqueeze(1)


# ------------------------------------------------------------------------------------------------ dec_qkv
@pytest.mark.parametrize("B,positions", [(1, [0]), (8, [1, 63, 64, 65, 5200, 6223, 32766, 127]), (9, [5200 + i for i in range(9)]),
                                         (16, [0, 1, 63, 64, 65, 127, 128, 5199, 5200, 5201, 6223, 6224, 12345, 32766, 31, 32]),
                                         (35, [(977 * i) % 7000 for i in range(35)])])                # three 16-row batch tiles, the last one ragged
def test_dec_qkv_norm_proj_bias_rope_and_page_append(eng, B, positions):
    g = torch.Generator().manual_seed(100 + B)
    h = bf(torch.randn(B, H, generator=g) * 2)
    ln_w = bf(1 + 0.1 * torch.randn(H, generator=g))
    W = bf(torch.randn((HQ + 2 * HKV) * 128, H, generator=g) * 0.02)
    bias = bf(torch.randn((HQ + 2 * HKV) * 128, generator=g) * 0.1)
    pos = torch.tensor(positions, dtype=torch.int64)
    max_pages = 512
    perm = torch.randperm(B + 3, generator=g)[:B]                       # a page table that is not the identity
    table = torch.full((B, max_pages), -1, dtype=torch.int32)
    for b in range(B):
        table[b, positions[b] >> 6] = int(perm[b])
    table.clamp_(min=0)
    sentinel = 0x7F7F                                                   # a bf16 NaN pattern nobody writes
    pool = torch.full((B + 3, HKV, 2, 8192), sentinel, dtype=torch.int16)
    pool_d = dev(pool)
    q_out = torch.zeros(B, HQ * 128, dtype=torch.bfloat16, device="cuda")
    hd, lnd, Wd_, bd = dev(h), dev(ln_w), dev(W), dev(bias)
    ctx_d, tab_d = dev(pos.to(torch.int32)), dev(table)
    torch.cuda.synchronize()
    eng.op_dec_qkv(hd.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), bd.data_ptr(), ctx_d.data_ptr(), tab_d.data_ptr(), max_pages,
                   pool_d.data_ptr(), q_out.data_ptr(), B, H, HQ, HKV, EPS, THETA)
    # ---- oracle (bf16-emulated)
    x = om.rms_norm(h.float(), ln_w.float(), EPS, True)
    qkv = om._r(om.linear(x, W.float(), bias.float()), True)
    q = qkv[:, :HQ * 128].view(B, HQ, 128)
    k = qkv[:, HQ * 128:(HQ + HKV) * 128].view(B, HKV, 128)
    v = qkv[:, (HQ + HKV) * 128:].view(B, HKV, 128)
    q_ref, k_ref = om._r(_rope(q, pos), True), om._r(_rope(k, pos), True)
    close(q_out.view(B, HQ, 128), q_ref, rel=2 ** -7, abs_=2e-3, what="q")
    got = pool_d.cpu()
    touched = torch.zeros_like(got, dtype=torch.bool)
    for b in range(B):
        pg, key = int(table[b, positions[b] >> 6]), positions[b] & 63
        for hk in range(HKV):
            kk = got[pg, hk, 0][K_IDX[key]].view(torch.bfloat16)
            vv = got[pg, hk, 1][V_IDX[key]].view(torch.bfloat16)
            close(kk, k_ref[b, hk], rel=2 ** -7, abs_=2e-3, what=f"k row {b}")
            close(vv, v[b, hk], rel=2 ** -7, abs_=2e-3, what=f"v row {b}")
            touched[pg, hk, 0][K_IDX[key]] = True
            touched[pg, hk, 1][V_IDX[key]] = True
    assert (got[~touched] == sentinel).all(), "the append wrote outside the new token's slots"


def test_dec_qkv_without_bias(eng):
    g = torch.Generator().manual_seed(7)
    B = 3
    h, ln_w = bf(torch.randn(B, H, generator=g)), bf(1 + 0.1 * torch.randn(H, generator=g))
    W = bf(torch.randn((HQ + 2 * HKV) * 128, H, generator=g) * 0.02)
    pos = torch.tensor([5, 64, 700])
    table = torch.arange(B, dtype=torch.int32).view(B, 1).expand(B, 16).contiguous()
    pool_d = torch.zeros(B, HKV, 2, 8192, dtype=torch.int16, device="cuda")
    q_out = torch.zeros(B, HQ * 128, dtype=torch.bfloat16, device="cuda")
    hd, lnd, Wd_, ctx_d, tab_d = dev(h), dev(ln_w), dev(W), dev(pos.to(torch.int32)), dev(table)
    torch.cuda.synchronize()
    eng.op_dec_qkv(hd.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), 0, ctx_d.data_ptr(), tab_d.data_ptr(), 16, pool_d.data_ptr(),
                   q_out.data_ptr(), B, H, HQ, HKV, EPS, THETA)
    qkv = om._r(om.linear(om.rms_norm(h.float(), ln_w.float(), EPS, True), W.float()), True)
    close(q_out.view(B, HQ, 128), om._r(_rope(qkv[:, :HQ * 128].view(B, HQ, 128), pos), True), abs_=2e-3, what="q")


# ------------------------------------------------------------------------------------------------ decode attention
def _attn_case(eng, ctxs, max_seq_len, seed, spike=False, plans=(None,)):
    """ctxs[b] = tokens already cached; the step attends over ctx + 1 keys (the appended token included).
    plans: dots_set_decode_plan values to run the kernel under (None = leave the engine as it is); every plan is held to the oracle and all
    plans to each other BIT FOR BIT (the streaming kernel of round 5 against the per-split kernel)."""
    B = len(ctxs)
    g = torch.Generator().manual_seed(seed)
    max_pages = (max_seq_len + 63) // 64
    n_pages = [(c + 1 + 63) // 64 for c in ctxs]
    total = sum(n_pages)
    perm = torch.randperm(total + 2, generator=g)
    table = torch.zeros(B, max_pages, dtype=torch.int32)
    pool = torch.zeros(total + 2, HKV, 2, 8192, dtype=torch.bfloat16)
    pool.view(torch.int16)[:] = 0x7F00                                        # 1.7e38 wherever nothing is packed: keys past ctx must be masked, not multiplied in
    q = bf(torch.randn(B, HQ, 128, generator=g))
    refs, off = [], 0
    for b, c in enumerate(ctxs):
        n = c + 1
        K = bf(torch.randn(n, HKV, 128, generator=g))
        Vv = bf(torch.randn(n, HKV, 128, generator=g))
        if spike and n > 200:
            K[n - 130] = bf(q[b, 0:HKV] * 6)                                  # one dominant key in a late page: running-max rescale path
        table[b, :n_pages[b]] = perm[off:off + n_pages[b]].to(torch.int32)
        off += n_pages[b]
        pack_pages(pool, table[b], K, Vv)
        rep = HQ // HKV
        ref = om._attention(q[b].float().unsqueeze(1), K.float().transpose(0, 1).repeat_interleave(rep, 0),
                            Vv.float().transpose(0, 1).repeat_interleave(rep, 0), 1 / math.sqrt(128), False, True)[:, 0]
        refs.append(ref)
    qd, pd, cd, td = dev(q.reshape(B, HQ * 128)), dev(pool), dev(torch.tensor(ctxs, dtype=torch.int32)), dev(table)
    outs = []
    try:
        for plan in plans:
            if plan is not None:
                eng.set_decode_plan(plan)
            out = torch.zeros(B, HQ * 128, dtype=torch.bfloat16, device="cuda")
            torch.cuda.synchronize()
            eng.op_decode_attn(qd.data_ptr(), pd.data_ptr(), cd.data_ptr(), td.data_ptr(), max_pages, out.data_ptr(), B, HQ, HKV, max_seq_len)
            got = out.view(B, HQ, 128).float().cpu()
            for b in range(B):
                close(got[b], refs[b], rel=2 ** -6, abs_=4e-3, what=f"plan {plan} seq {b} ctx {ctxs[b]}")
            outs.append(out.cpu().view(torch.int16))
    finally:
        if any(pl is not None for pl in plans):
            eng.set_decode_plan(0)
    for plan, o in zip(plans[1:], outs[1:]):
        assert torch.equal(o, outs[0]), f"decode attention under plan {plan} differs from plan {plans[0]} in {int((o != outs[0]).sum())} elements"


@pytest.mark.parametrize("ctxs,max_seq_len", [
    ([0], 64),                                                   # one key, one split
    ([1, 63, 64, 65, 0, 127, 128, 255], 640),                    # page edges, 3 splits
    ([5200] * 8, 6224),                                          # the bench: 8 rows, 25 splits
    ([1, 63, 64, 65, 5200, 6223, 5199, 300], 6224),              # ragged batch, empty splits for the short rows
    ([5200 + 100 * i for i in range(9)], 6224),                  # B = 9: 16-row X image
    ([6223] * 16, 6224),                                         # B = 16, every page full but the last key
    ([32766, 5200, 64], 32768),                                  # 64 splits, 2 page rounds per wave for the long row
    ([(613 * i) % 1500 + 1 for i in range(21)], 1600),           # B = 21: the combine writes two X-image tiles
])
def test_decode_attention_matches_oracle(eng, ctxs, max_seq_len):
    _attn_case(eng, ctxs, max_seq_len, seed=sum(ctxs) + len(ctxs))


@pytest.mark.parametrize("ctxs,max_seq_len", [
    ([0], 64),                                                   # one item, one wave with a page
    ([1, 63, 64, 65, 0, 127, 128, 255], 640),                    # page edges: waves without a page inside an active split
    ([5200] * 8, 6224),                                          # the bench's rows
    ([1, 63, 64, 65, 5200, 6223, 5199, 300], 6224),              # ragged: most splits of the short rows are skipped by the walk
    ([5150 + (37 * i) % 1000 for i in range(64)], 6288),         # the pipelined step: 64 rows x 25 splits = 3200 items on the resident workgroups
    ([(613 * i) % 1500 + 1 for i in range(21)], 1600),           # B = 21: the combine writes two X-image tiles
    ([16383, 64, 9000], 16384),                                  # 64 splits x 4 waves = 256 pages: the largest capacity the one-page-per-wave plan covers
])
def test_decode_attention_stream_equals_per_split_bitwise(eng, ctxs, max_seq_len):
    """Round 5: the streaming kernel (one resident workgroup per CU walking the items, pages by LDS-DMA one item ahead) against the per-split
    kernel: plan 4 = per split, 2 = streaming on the whole chip, 3 = streaming on the partition plan's workgroup count.  Same bits."""
    _attn_case(eng, ctxs, max_seq_len, seed=sum(ctxs) + len(ctxs), plans=(4, 2, 3))


def test_decode_attention_spiked_key_stream(eng):
    _attn_case(eng, [5200, 777, 6000], 6224, seed=5, spike=True, plans=(4, 2))


def test_decode_attention_dominant_late_key(eng):
    _attn_case(eng, [5200, 900, 6000], 6224, seed=5, spike=True)


# ------------------------------------------------------------------------------------------------ projections
@pytest.mark.parametrize("B", [1, 8, 9, 16, 17, 40])      # 17 / 40: two and three 16-row batch tiles
@pytest.mark.parametrize("N,K", [(H, HQ * 128), (H, I)])
def test_dec_proj_residual(eng, B, N, K):
    g = torch.Generator().manual_seed(B * 31 + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    h = bf(torch.randn(B, N, generator=g) * 2)
    hd, xd, Wd_ = dev(h.clone()), dev(x), dev(W)
    torch.cuda.synchronize()
    eng.op_dec_proj(xd.data_ptr(), Wd_.data_ptr(), hd.data_ptr(), B, N, K)
    ref = om._r(h.float() + x.float() @ W.float().t(), True)
    close(hd, ref, what=f"proj N={N} K={K}")
    if B > 1:                                   # transpose / row-mixing detection: rows are not interchangeable
        assert (hd.float().cpu() - ref.flip(0)).abs().max() > 0.1


@pytest.mark.parametrize("B", [1, 8, 9, 16, 17, 40])      # 17 / 40: two and three 16-row batch tiles
def test_dec_gateup_swiglu(eng, B):
    g = torch.Generator().manual_seed(B + 77)
    h, ln_w = bf(torch.randn(B, H, generator=g) * 3), bf(1 + 0.1 * torch.randn(H, generator=g))
    gate, up = bf(torch.randn(I, H, generator=g) / math.sqrt(H)), bf(torch.randn(I, H, generator=g) / math.sqrt(H))
    out = torch.zeros(B, I, dtype=torch.bfloat16, device="cuda")
    hd, lnd, gd, ud = dev(h), dev(ln_w), dev(gate), dev(up)
    torch.cuda.synchronize()
    eng.op_dec_gateup(hd.data_ptr(), lnd.data_ptr(), gd.data_ptr(), ud.data_ptr(), out.data_ptr(), B, H, I, EPS)
    x = om.rms_norm(h.float(), ln_w.float(), EPS, True)
    ref = om._r(torch.nn.functional.silu(x @ gate.float().t()) * (x @ up.float().t()), True)
    close(out, ref, rel=2 ** -6, abs_=2e-3, what="gate/up")


@pytest.mark.parametrize("B", [1, 8, 9, 16, 17, 40])      # 17 / 40: two and three 16-row batch tiles
def test_dec_lmhead_logits(eng, B):
    g = torch.Generator().manual_seed(B + 5)
    h, ln_w = bf(torch.randn(B, H, generator=g) * 3), bf(1 + 0.1 * torch.randn(H, generator=g))
    W = bf(torch.randn(V, H, generator=g) * 0.02)
    out = torch.zeros(B, V, dtype=torch.float32, device="cuda")
    hd, lnd, Wd_ = dev(h), dev(ln_w), dev(W)
    torch.cuda.synchronize()
    eng.op_dec_lmhead(hd.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), out.data_ptr(), B, H, V, EPS)
    ref = om.rms_norm(h.float(), ln_w.float(), EPS, True) @ W.float().t()
    close(out, ref, rel=1e-4, abs_=1e-3, what="logits")          # a 1-ulp difference in rsqrt flips a few bf16 roundings of X: ~1e-3 on a range of 8
    assert torch.equal(out.argmax(-1).cpu(), ref.argmax(-1)) or (ref.topk(2).values.diff().abs().min() < 1e-3)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B", [17, 33, 40, 64])
def test_two_tile_kernels_equal_the_one_tile_kernels_bitwise(eng, B, fp8):
    """Above 16 rows gate|up and lm_head run TWO 16-row batch tiles per workgroup (the weights cross the CU once per 32 rows).  Per output
    element the MFMAs, their order and the split-K reduction are those of the one-tile kernels, so the rows of a B-row call must equal,
    bit for bit, the same rows computed in calls of at most 16 rows — the property batch invariance of the tokens rests on.  fp8: the e4m3
    instantiations of the same templates (weights quantised inside the op entry points, quant.hip)."""
    g = torch.Generator().manual_seed(B + 1234)
    h, ln_w = bf(torch.randn(B, H, generator=g) * 3), bf(1 + 0.1 * torch.randn(H, generator=g))
    gate, up = bf(torch.randn(I, H, generator=g) / math.sqrt(H)), bf(torch.randn(I, H, generator=g) / math.sqrt(H))
    W = bf(torch.randn(V, H, generator=g) * 0.02)
    hd, lnd, gd, ud, Wd_ = dev(h), dev(ln_w), dev(gate), dev(up), dev(W)
    act = torch.zeros(B, I, dtype=torch.bfloat16, device="cuda")
    logits = torch.zeros(B, V, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.op_dec_gateup(hd.data_ptr(), lnd.data_ptr(), gd.data_ptr(), ud.data_ptr(), act.data_ptr(), B, H, I, EPS, fp8=fp8)
    eng.op_dec_lmhead(hd.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), logits.data_ptr(), B, H, V, EPS, fp8=fp8)
    eng.synchronize()
    assert float(act.float().abs().max()) > 0 and float(logits.abs().max()) > 0
    for r0 in range(0, B, 16):
        n = min(16, B - r0)
        hs = hd[r0:r0 + n].contiguous()
        a1 = torch.zeros(n, I, dtype=torch.bfloat16, device="cuda")
        l1 = torch.zeros(n, V, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        eng.op_dec_gateup(hs.data_ptr(), lnd.data_ptr(), gd.data_ptr(), ud.data_ptr(), a1.data_ptr(), n, H, I, EPS, fp8=fp8)
        eng.op_dec_lmhead(hs.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), l1.data_ptr(), n, H, V, EPS, fp8=fp8)
        eng.synchronize()
        assert torch.equal(act[r0:r0 + n].view(torch.int16), a1.view(torch.int16)), f"gate|up rows {r0}..{r0 + n - 1} of {B}"
        assert torch.equal(logits[r0:r0 + n].view(torch.int32), l1.view(torch.int32)), f"lm_head rows {r0}..{r0 + n - 1} of {B}"


@pytest.mark.parametrize("cus", [None, 96, 64, 32])        # whole chip: 4 waves, deep ring; 96 CUs: 8 waves; 64: 12 waves, one round; 32: 12 waves, two rounds of gate|up
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B", [33, 40, 64])
def test_four_tile_kernels_equal_the_one_tile_kernels_bitwise(eng, B, fp8, cus, monkeypatch):
    """Round 6 (csrc/decode_b64.hip): above 32 rows gate|up and lm_head run ALL FOUR batch tiles in one workgroup — a norm kernel writes the
    X image once, a wave owns a pair of weight tiles for all rows and walks K in order through an LDS ring of X chunks.  Per output element
    the chains are the per-tile kernels' (four K slices summed in order / even + odd k-steps), so the rows of a B-row call must equal, bit
    for bit, the same rows computed in calls of at most 16 rows, for every launch shape (waves per workgroup, ring depth, rounds)."""
    if cus is None:
        monkeypatch.delenv("DOTS_OCR_DEC_WIDE_CUS", raising=False)
    else:
        monkeypatch.setenv("DOTS_OCR_DEC_WIDE_CUS", str(cus))
    g = torch.Generator().manual_seed(B * 13 + 7)
    h, ln_w = bf(torch.randn(B, H, generator=g) * 3), bf(1 + 0.1 * torch.randn(H, generator=g))
    gate, up = bf(torch.randn(I, H, generator=g) / math.sqrt(H)), bf(torch.randn(I, H, generator=g) / math.sqrt(H))
    W = bf(torch.randn(V, H, generator=g) * 0.02)
    hd, lnd, gd, ud, Wd_ = dev(h), dev(ln_w), dev(gate), dev(up), dev(W)
    act = torch.zeros(B, I, dtype=torch.bfloat16, device="cuda")
    logits = torch.zeros(B, V, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.op_dec_gateup(hd.data_ptr(), lnd.data_ptr(), gd.data_ptr(), ud.data_ptr(), act.data_ptr(), B, H, I, EPS, fp8=fp8)
    eng.op_dec_lmhead(hd.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), logits.data_ptr(), B, H, V, EPS, fp8=fp8)
    eng.synchronize()
    assert float(act.float().abs().max()) > 0 and float(logits.abs().max()) > 0
    assert torch.isfinite(act.float()).all() and torch.isfinite(logits).all()
    monkeypatch.delenv("DOTS_OCR_DEC_WIDE_CUS", raising=False)
    for r0 in range(0, B, 16):
        n = min(16, B - r0)
        hs = hd[r0:r0 + n].contiguous()
        a1 = torch.zeros(n, I, dtype=torch.bfloat16, device="cuda")
        l1 = torch.zeros(n, V, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        eng.op_dec_gateup(hs.data_ptr(), lnd.data_ptr(), gd.data_ptr(), ud.data_ptr(), a1.data_ptr(), n, H, I, EPS, fp8=fp8)
        eng.op_dec_lmhead(hs.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), l1.data_ptr(), n, H, V, EPS, fp8=fp8)
        eng.synchronize()
        assert torch.equal(act[r0:r0 + n].view(torch.int16), a1.view(torch.int16)), f"gate|up rows {r0}..{r0 + n - 1} of {B}"
        assert torch.equal(logits[r0:r0 + n].view(torch.int32), l1.view(torch.int32)), f"lm_head rows {r0}..{r0 + n - 1} of {B}"
    if cus is None and not fp8:       # and against the oracle (the <= 16-row kernels are held to it above; this is the same check on the new path)
        ref = om.rms_norm(h.float(), ln_w.float(), EPS, True) @ W.float().t()
        close(logits, ref, rel=1e-4, abs_=1e-3, what="logits of the four-tile lm_head")


@pytest.mark.parametrize("cus", [None, 100, 64, 48, 128, 77, 400])     # CUs the launcher plans for: whole chip (1 unit / 1 tile per workgroup), 2 units, 3 units / 2 tiles, 4 units;
@pytest.mark.parametrize("fp8", [False, True])                          # K-split projections (round 6, above 32 rows): 3 / 8 / 12 / 12 / 6 / 10 / 2 units = 2 / 4 / 6 / 6 / 3 / 5 / 1 MFMAs per k-step
@pytest.mark.parametrize("B", [17, 33, 40, 64])
def test_wide_kernels_equal_the_one_tile_kernels_bitwise(eng, B, fp8, cus, monkeypatch):
    """Round 6 addition: above 32 rows the projections run as FOUR K quarters (dec_proj_ksplit_kernel, decode_b64.hip: a quarter of the X image per workgroup,
    the quarters' fp32 sums added by the consumer's norm kernel — here by the single-kernel entry point's residual-update launch) — the same bitwise contract,
    for every number of MFMAs per k-step the launcher can choose.
    Round 5: above 16 rows dec_qkv and the two projections run the WIDE kernels (all batch tiles in one workgroup, no LDS X image, parity
    passes over the K slices; csrc/decode_fused.hip).  Per output element their arithmetic is the per-tile kernels': the rows of a B-row
    call equal, bit for bit, the same rows computed in calls of at most 16 rows — q, the appended K / V page slots, and both residual
    projections (o_proj K = 1536, down_proj K = 8960) — for every workgroup shape the launcher can choose (DOTS_OCR_DEC_WIDE_CUS)."""
    if cus is None:
        monkeypatch.delenv("DOTS_OCR_DEC_WIDE_CUS", raising=False)
    else:
        monkeypatch.setenv("DOTS_OCR_DEC_WIDE_CUS", str(cus))
    monkeypatch.setenv("DOTS_OCR_DEC_KSPLIT", "2")          # both projections through the K-split kernel above 32 rows (the default splits down_proj only)
    g = torch.Generator().manual_seed(B * 7 + 5)
    h = bf(torch.randn(B, H, generator=g) * 2)
    ln_w = bf(1 + 0.1 * torch.randn(H, generator=g))
    W = bf(torch.randn((HQ + 2 * HKV) * 128, H, generator=g) * 0.02)
    bias = bf(torch.randn((HQ + 2 * HKV) * 128, generator=g) * 0.1)
    positions = [(977 * i + 63) % 7000 for i in range(B)]
    pos = torch.tensor(positions, dtype=torch.int32)
    max_pages = 128
    perm = torch.randperm(B + 3, generator=g)[:B]
    table = torch.zeros((B, max_pages), dtype=torch.int32)
    for b in range(B):
        table[b, positions[b] >> 6] = int(perm[b])
    sentinel = 0x7F7F
    hd, lnd, Wd_, bd, ctx_d, tab_d = dev(h), dev(ln_w), dev(W), dev(bias), dev(pos), dev(table)

    def qkv(r0, n):
        pool = dev(torch.full((B + 3, HKV, 2, 8192), sentinel, dtype=torch.int16))
        q = torch.zeros(n, HQ * 128, dtype=torch.bfloat16, device="cuda")
        hs, cs, ts = hd[r0:r0 + n].contiguous(), ctx_d[r0:r0 + n].contiguous(), tab_d[r0:r0 + n].contiguous()
        torch.cuda.synchronize()
        eng.op_dec_qkv(hs.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), bd.data_ptr(), cs.data_ptr(), ts.data_ptr(), max_pages, pool.data_ptr(),
                       q.data_ptr(), n, H, HQ, HKV, EPS, THETA, fp8=fp8)
        eng.synchronize()
        return q, pool

    q_all, pool_all = qkv(0, B)
    assert float(q_all.float().abs().max()) > 0
    pool_tiles = torch.full_like(pool_all, sentinel)
    for r0 in range(0, B, 16):
        n = min(16, B - r0)
        q1, p1 = qkv(r0, n)
        assert torch.equal(q_all[r0:r0 + n].view(torch.int16), q1.view(torch.int16)), f"q rows {r0}..{r0 + n - 1} of {B}"
        wrote = p1 != sentinel
        assert not (wrote & (pool_tiles != sentinel)).any()
        pool_tiles[wrote] = p1[wrote]
    assert torch.equal(pool_all, pool_tiles), f"K / V page slots of a {B}-row call differ from the per-tile calls"

    for N, K in [(H, HQ * 128), (H, I)]:
        x = bf(torch.randn(B, K, generator=g))
        Wp = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
        r = bf(torch.randn(B, N, generator=g) * 2)
        xd, Wpd = dev(x), dev(Wp)
        h_all = dev(r.clone())
        torch.cuda.synchronize()
        eng.op_dec_proj(xd.data_ptr(), Wpd.data_ptr(), h_all.data_ptr(), B, N, K, fp8=fp8)
        eng.synchronize()
        assert not torch.equal(h_all.cpu(), r)
        for r0 in range(0, B, 16):
            n = min(16, B - r0)
            h1, xs = dev(r[r0:r0 + n].clone()), xd[r0:r0 + n].contiguous()
            torch.cuda.synchronize()
            eng.op_dec_proj(xs.data_ptr(), Wpd.data_ptr(), h1.data_ptr(), n, N, K, fp8=fp8)
            eng.synchronize()
            assert torch.equal(h_all[r0:r0 + n].view(torch.int16), h1.view(torch.int16)), f"proj K={K} rows {r0}..{r0 + n - 1} of {B}"


def test_decode_kernels_reject_unsupported_shapes(eng):
    from dots_ocr_amd.engine import DotsEngineError
    z = torch.zeros(65 * 2048, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(DotsEngineError):
        eng.op_dec_proj(z.data_ptr(), z.data_ptr(), z.data_ptr(), 65, 64, 512)          # B > 64 (four 16-row tiles)
    with pytest.raises(DotsEngineError):
        eng.op_dec_gateup(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 2, 2048, 64, EPS)   # hidden > 1536
