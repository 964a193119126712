"""fp8 configuration (BASELINE configs[4]; DotsConfig.fp8_weights, csrc/quant.hip): OCP e4m3 weights with one fp32 scale per output
channel — streamed as bytes against bf16 activations by the decode kernels (W8A16), multiplied with per-token e4m3 activations on
the fp8 MFMA by the ViT / prefill GEMMs (W8A8, gemm.hip: gemm_fp8_256pp_kernel).

  * the quantiser equals torch's float8_e4m3fn cast bit for bit (values and scales), zero rows and tiny rows included;
  * every decode kernel's fp8 instantiation, at the BASELINE dimensions, equals the oracle run on the quantised weights
    (oracle/model.py quantize_rows_fp8: the "fp8 oracle mode") to the same tolerances as the bf16 kernels;
  * the fp8-MFMA GEMM (activation quantiser + kernel + row/column-scale epilogues) equals the oracle's a8 linear; the bf16 GEMM's
    per-column scale epilogue equals the oracle on the quantised weights;
  * a whole fp8 engine (ViT -> prefill -> decode) follows the oracle's fp8 mode (quantize_fp8_state_dict + fp8_act=True).
"""
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


def dq(w):
    """bf16 weight -> the fp32 matrix q * scale the fp8 configuration computes with."""
    q, s = om.quantize_rows_fp8(w)
    return q * s[:, None]


# KV page layout (csrc/decode.hip header) and the rope convention, as in tests/test_decode_kernels_gpu.py
_key, _d = torch.arange(64).view(64, 1), torch.arange(128).view(1, 128)
K_IDX = (((_key >> 4) * 4 + (_d >> 5)) * 64 + ((_d >> 3) & 3) * 16 + (_key & 15)) * 8 + (_d & 7)
V_IDX = (((_key >> 5) * 8 + (_d >> 4)) * 64 + (((_key & 31) >> 2) & 3) * 16 + (_d & 15)) * 8 + 4 * ((_key & 31) >> 4) + (_key & 3)


def _rope(x, pos):
    cos, sin = om.lm_rope_cos_sin(pos, 128, THETA)
    return x * cos.unsqueeze(1) + om.rotate_half(x) * sin.unsqueeze(1)


# ------------------------------------------------------------------------------------------------ quantiser
def test_quantiser_equals_torch_e4m3_cast_bit_for_bit(eng):
    g = torch.Generator().manual_seed(5)
    N, K = 517, 1536
    w = torch.randn(N, K, generator=g) * torch.logspace(-6, 2, N).view(N, 1)        # row magnitudes over 8 decades
    w[7] = 0                                                                         # all-zero row -> scale 1, q 0
    w[11, 1:] *= 1e-4                                                                # one dominant element: the rest lands in e4m3 subnormals / zero
    w[13] = torch.linspace(-1, 1, K)                                                 # every binade of the format, ties included
    w = bf(w)
    wd = dev(w.clone())
    sc = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.op_quant_fp8(wd.data_ptr(), sc.data_ptr(), N, K)
    q_ref, s_ref = om.quantize_rows_fp8(w)
    assert torch.equal(sc.cpu().view(torch.int32), s_ref.view(torch.int32)), "scales differ"
    got = wd.cpu().float()
    assert torch.equal(got, q_ref), f"{int((got != q_ref).sum())} quantised values differ from torch's float8_e4m3fn cast"
    assert torch.equal(bf(q_ref).float(), q_ref)                                     # every e4m3 value is a bf16 value


# ------------------------------------------------------------------------------------------------ decode kernels
@pytest.mark.parametrize("B", [1, 8, 16])
def test_fp8_dec_qkv(eng, B):
    g = torch.Generator().manual_seed(200 + B)
    h = bf(torch.randn(B, H, generator=g) * 2)
    ln_w = bf(1 + 0.1 * torch.randn(H, generator=g))
    W = bf(torch.randn((HQ + 2 * HKV) * 128, H, generator=g) * 0.02 * torch.logspace(-1, 1, (HQ + 2 * HKV) * 128).view(-1, 1))
    bias = bf(torch.randn((HQ + 2 * HKV) * 128, generator=g) * 0.1)
    positions = [(37 * b * b + 5200 * (b & 1)) % 6000 for b in range(B)]
    pos = torch.tensor(positions, dtype=torch.int64)
    max_pages = 128
    table = torch.zeros((B, max_pages), dtype=torch.int32)
    for b in range(B):
        table[b, positions[b] >> 6] = B - 1 - b
    pool_d = torch.zeros(B, HKV, 2, 8192, dtype=torch.bfloat16, device="cuda")
    q_out = torch.zeros(B, HQ * 128, dtype=torch.bfloat16, device="cuda")
    hd, lnd, Wd_, bd = dev(h), dev(ln_w), dev(W), dev(bias)
    ctx_d, tab_d = dev(pos.to(torch.int32)), dev(table)
    torch.cuda.synchronize()
    eng.op_dec_qkv(hd.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), bd.data_ptr(), ctx_d.data_ptr(), tab_d.data_ptr(), max_pages,
                   pool_d.data_ptr(), q_out.data_ptr(), B, H, HQ, HKV, EPS, THETA, fp8=True)
    assert torch.equal(Wd_.cpu(), W), "the op quantises a copy, not the caller's weight"
    x = om.rms_norm(h.float(), ln_w.float(), EPS, True)
    qkv = om._r(om.linear(x, dq(W), bias.float()), True)
    q = qkv[:, :HQ * 128].view(B, HQ, 128)
    k = qkv[:, HQ * 128:(HQ + HKV) * 128].view(B, HKV, 128)
    v = qkv[:, (HQ + HKV) * 128:].view(B, HKV, 128)
    q_ref, k_ref = om._r(_rope(q, pos), True), om._r(_rope(k, pos), True)
    close(q_out.view(B, HQ, 128), q_ref, rel=2 ** -7, abs_=2e-3, what="q")
    got = pool_d.cpu()
    for b in range(B):
        pg, key = int(table[b, positions[b] >> 6]), positions[b] & 63
        for hk in range(HKV):
            close(got[pg, hk, 0][K_IDX[key]], k_ref[b, hk], rel=2 ** -7, abs_=2e-3, what=f"k row {b}")
            close(got[pg, hk, 1][V_IDX[key]], v[b, hk], rel=2 ** -7, abs_=2e-3, what=f"v row {b}")


@pytest.mark.parametrize("B,N,K", [(1, H, HQ * 128), (8, H, I), (16, H, I), (9, H, HQ * 128)])
def test_fp8_dec_proj(eng, B, N, K):
    g = torch.Generator().manual_seed(300 + B + K)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.02 * torch.logspace(-1, 1, N).view(-1, 1))
    h = bf(torch.randn(B, N, generator=g))
    hd = dev(h.clone())
    xd, Wd_ = dev(x), dev(W)
    torch.cuda.synchronize()
    eng.op_dec_proj(xd.data_ptr(), Wd_.data_ptr(), hd.data_ptr(), B, N, K, fp8=True)
    ref = om._r(h.float() + om.linear(x.float(), dq(W)), True)
    close(hd, ref, rel=2 ** -7, abs_=2e-3, what="proj")


@pytest.mark.parametrize("B", [1, 8, 16])
def test_fp8_dec_gateup(eng, B):
    g = torch.Generator().manual_seed(400 + B)
    h = bf(torch.randn(B, H, generator=g) * 2)
    ln_w = bf(1 + 0.1 * torch.randn(H, generator=g))
    Wg = bf(torch.randn(I, H, generator=g) * 0.03 * torch.logspace(-0.5, 0.5, I).view(-1, 1))
    Wu = bf(torch.randn(I, H, generator=g) * 0.03 * torch.logspace(0.5, -0.5, I).view(-1, 1))
    act = torch.zeros(B, I, dtype=torch.bfloat16, device="cuda")
    hd, lnd, gd, ud = dev(h), dev(ln_w), dev(Wg), dev(Wu)
    torch.cuda.synchronize()
    eng.op_dec_gateup(hd.data_ptr(), lnd.data_ptr(), gd.data_ptr(), ud.data_ptr(), act.data_ptr(), B, H, I, EPS, fp8=True)
    x = om.rms_norm(h.float(), ln_w.float(), EPS, True)
    ref = om._r(torch.nn.functional.silu(om.linear(x, dq(Wg))) * om.linear(x, dq(Wu)), True)
    close(act, ref, rel=2 ** -6, abs_=2e-3, what="act")


@pytest.mark.parametrize("B", [1, 8, 16])
def test_fp8_dec_lmhead(eng, B):
    g = torch.Generator().manual_seed(500 + B)
    h = bf(torch.randn(B, H, generator=g) * 3)
    ln_w = bf(1 + 0.1 * torch.randn(H, generator=g))
    W = bf(torch.randn(V, H, generator=g) * 0.02)
    logits = torch.zeros(B, V, dtype=torch.float32, device="cuda")
    hd, lnd, Wd_ = dev(h), dev(ln_w), dev(W)
    torch.cuda.synchronize()
    eng.op_dec_lmhead(hd.data_ptr(), lnd.data_ptr(), Wd_.data_ptr(), logits.data_ptr(), B, H, V, EPS, fp8=True)
    x = om.rms_norm(h.float(), ln_w.float(), EPS, True)
    ref = om.linear(x, dq(W))
    close(logits, ref, rel=0, abs_=1e-3, what="logits")
    assert torch.equal(logits.cpu().argmax(-1), ref.argmax(-1)) or float((ref.topk(2).values[:, 0] - ref.topk(2).values[:, 1]).min()) < 1e-2


# ------------------------------------------------------------------------------------------------ GEMM column scale
@pytest.mark.parametrize("epi", ["none", "residual", "swiglu", "gelu"])
def test_gemm_column_scale_epilogue(eng, epi):
    from dots_ocr_amd.engine import EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU
    g = torch.Generator().manual_seed(17)
    M, N, K = 300, 512, 768
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05 * torch.logspace(-1, 1, N).view(-1, 1))
    bias = bf(torch.randn(N, generator=g) * 0.1)
    R = bf(torch.randn(M, N, generator=g))
    Wq = dev(W.clone())
    sc = torch.zeros(N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.op_quant_fp8(Wq.data_ptr(), sc.data_ptr(), N, K)
    Ad, bd, Rd = dev(A), dev(bias), dev(R)
    Wdq = dq(W)
    if epi == "swiglu":
        # packed rows: 64-row groups = 32 gate rows | 32 up rows; the scale index is the packed row
        Cd = torch.zeros(M, N // 2, dtype=torch.bfloat16, device="cuda")
        eng.op_gemm(Ad.data_ptr(), Wq.data_ptr(), bd.data_ptr(), None, Cd.data_ptr(), M, N, K, EPI_SWIGLU, sc.data_ptr())
        y = (om.linear(A.float(), Wdq, bias.float())).view(M, N // 64, 2, 32)
        ref = om._r(torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1], True).reshape(M, N // 2)
    else:
        Cd = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        code = {"none": EPI_NONE, "residual": EPI_RESIDUAL, "gelu": EPI_GELU}[epi]
        eng.op_gemm(Ad.data_ptr(), Wq.data_ptr(), bd.data_ptr(), Rd.data_ptr() if epi == "residual" else None, Cd.data_ptr(), M, N, K, code,
                    sc.data_ptr())
        y = om.linear(A.float(), Wdq, bias.float())
        if epi == "residual":
            y = y + R.float()
        if epi == "gelu":
            y = torch.nn.functional.gelu(y)
        ref = om._r(y, True)
    eng.synchronize()
    close(Cd, ref, rel=2 ** -7, abs_=2e-3, what=epi)


@pytest.mark.parametrize("M,N,K", [(300, 256, 64), (1000, 768, 192), (4100, 1536, 1536), (515, 512, 4224)])
@pytest.mark.parametrize("epi", ["none", "residual", "gelu"])
def test_fp8_mfma_gemm_equals_the_oracle_a8_linear(eng, M, N, K, epi):
    from dots_ocr_amd.engine import EPI_GELU, EPI_NONE, EPI_RESIDUAL
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g) * torch.logspace(-2, 1, M).view(-1, 1))                # per-token scales over 3 decades
    A[3] = 0                                                                                       # an all-zero token
    W = bf(torch.randn(N, K, generator=g) / K ** 0.5 * torch.logspace(-1, 1, N).view(-1, 1))
    bias = bf(torch.randn(N, generator=g) * 0.1)
    R = bf(torch.randn(M, N, generator=g))
    Ad, Wd_, bd, Rd = dev(A), dev(W), dev(bias), dev(R)
    Cd = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    code = {"none": EPI_NONE, "residual": EPI_RESIDUAL, "gelu": EPI_GELU}[epi]
    eng.op_gemm_fp8(Ad.data_ptr(), Wd_.data_ptr(), bd.data_ptr(), Rd.data_ptr() if epi == "residual" else None, Cd.data_ptr(), M, N, K, code)
    y = om.linear(A.float(), dq(W), bias.float(), a8=True)
    if epi == "residual":
        y = y + R.float()
    if epi == "gelu":
        y = torch.nn.functional.gelu(y)
    close(Cd, om._r(y, True), rel=2 ** -7, abs_=2e-3, what=f"fp8 gemm {epi}")
    assert torch.equal(Ad.cpu(), A) and torch.equal(Wd_.cpu(), W), "the op quantises copies"


def test_fp8_mfma_gemm_swiglu_and_repeatability(eng):
    from dots_ocr_amd.engine import EPI_SWIGLU
    g = torch.Generator().manual_seed(77)
    M, I, K = 3000, 1024, 1536
    A = bf(torch.randn(M, K, generator=g))
    Wg, Wu = bf(torch.randn(I, K, generator=g) / K ** 0.5), bf(torch.randn(I, K, generator=g) / K ** 0.5 * 2)
    W13 = torch.stack([Wg.view(I // 32, 32, K), Wu.view(I // 32, 32, K)], dim=1).reshape(2 * I, K).contiguous()
    Ad, Wd_ = dev(A), dev(W13)
    outs = []
    for _ in range(4):
        Cd = torch.zeros(M, I, dtype=torch.bfloat16, device="cuda")
        eng.op_gemm_fp8(Ad.data_ptr(), Wd_.data_ptr(), None, None, Cd.data_ptr(), M, 2 * I, K, EPI_SWIGLU)
        outs.append(Cd.cpu())
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16)), "fp8 GEMM result changes between runs"
    ref = om._r(torch.nn.functional.silu(om.linear(A.float(), dq(Wg), None, a8=True)) * om.linear(A.float(), dq(Wu), None, a8=True), True)
    close(outs[0], ref, rel=2 ** -6, abs_=2e-3, what="fp8 swiglu")


# ------------------------------------------------------------------------------------------------ whole engine
def test_fp8_engine_follows_the_oracle_on_the_quantised_state_dict():
    """Small dimensions, both phases: ViT + prefill (per-token e4m3 activations x e4m3 weights on the fp8 MFMA) and 12 decode steps
    (e4m3 byte stream x bf16 activations), greedy, teacher-forced oracle on quantize_fp8_state_dict(sd) with fp8_act=True.  The
    fp8 model is NOT the bf16 model: the same prompt through a bf16 engine must give different logits."""
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.image_utils import preprocess_image
    from dots_ocr_amd.synthetic import synth_page, synth_prompt_ids
    from dots_ocr_amd.weights import random_state_dict
    cfg = DotsConfig.tiny(layers=3, v_layers=2)
    sd = random_state_dict(cfg, seed=3)
    pv, thw = preprocess_image(synth_page(1, (224, 196)))
    ids = synth_prompt_ids(cfg, pv.shape[0] // 4, seed=1)
    n_new = 12
    out = {}
    for fp8 in (True, False):
        e = Engine(cfg, max_batch=1, max_seq_len=1024, max_patches=1024, fp8_weights=fp8)
        e.load_state_dict(sd)
        e.vit_forward(pv, np.asarray([thw], np.int64))
        e.prefill(ids, np.asarray([len(ids)], np.int32))
        lg, tk = [e.get_logits()[0].copy()], [int(e.get_last_tokens()[0])]
        for _ in range(1, n_new):
            e.decode_step()
            lg.append(e.get_logits()[0].copy())
            tk.append(int(e.get_last_tokens()[0]))
        e.close()
        out[fp8] = (lg, tk)
    lg, tk = out[True]
    qsd = om.quantize_fp8_state_dict(sd)
    _, ref = om.generate(qsd, cfg, torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(pv), torch.tensor([thw]), n_new,
                         emulate_bf16=True, forced_tokens=tk, return_logits=True, fp8_act=True)
    _, ref32 = om.generate(qsd, cfg, torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(pv), torch.tensor([thw]), n_new,
                           emulate_bf16=False, forced_tokens=tk, return_logits=True, fp8_act=True)
    # Tolerances: per-token activation quantisation is a step function — a bf16-level difference between two computations of the same
    # activation can flip an e4m3 rounding (6 % of that element), so W8A8 logits scatter more around the oracle than the bf16
    # configuration's (3 % of the logit range there): 4 % vs the emulated oracle, 6 % vs the fp32 one.
    worst, worst_emu = 0.0, 0.0
    for s in range(n_new):
        got = torch.from_numpy(lg[s])
        rng = float(ref32[s].max() - ref32[s].min())
        worst = max(worst, float((got - ref32[s]).abs().max()) / rng)
        worst_emu = max(worst_emu, float((got - ref[s]).abs().max()) / rng)
        best = int(ref[s].argmax())
        assert tk[s] == best or float(ref[s][best] - ref[s][tk[s]]) < 0.05 * rng, f"step {s}: token {tk[s]} vs fp8 oracle {best}"
    print(f"fp8 engine: max |logit err| {worst_emu:.4f} (emulated oracle) / {worst:.4f} (fp32 oracle) of the logit range")
    assert worst_emu < 0.04 and worst < 0.06, f"max |logit err| = {worst_emu:.4f} (emulated) / {worst:.4f} (fp32) of the logit range"
    # and the quantisation is really in effect
    d = max(float(np.abs(a - b).max()) for a, b in zip(out[True][0], out[False][0]))
    assert d > 1e-2, "fp8 and bf16 engines produced the same logits"
