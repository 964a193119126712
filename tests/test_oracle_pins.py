"""Pin the CPU oracle before trusting it (CPU only).

* LM restatement  == in-container transformers Qwen2ForCausalLM (the reference's dependency).
* image processor == in-container transformers Qwen2VLImageProcessorPil.
* smart_resize    == golden answers generated from the reference function.
* vision building blocks (2-D rope ids, rotary application, PatchMerger) == transformers analogues.
The vision tower as a whole is unpinned (its source is HF-hub remote code that is not available).
"""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from dots_ocr_amd.config import DotsConfig
from dots_ocr_amd.weights import random_state_dict
from oracle import image_processor as oip
from oracle import model as om

GOLD = Path(__file__).parent / "golden"


def test_smart_resize_golden():
    cases = json.loads((GOLD / "smart_resize.json").read_text())
    assert len(cases) > 500
    for c in cases:
        try:
            got = list(oip.smart_resize(c["h"], c["w"], 28, c["min_pixels"], c["max_pixels"]))
        except ValueError:
            got = "ValueError"
        assert got == c["out"], c


def test_smart_resize_survey_known_answers():
    # SURVEY §8(c), computed with the reference function
    for (h, w), exp in {(2250, 1700): (2240, 1708), (2339, 1654): (2352, 1652), (1344, 1344): (1344, 1344),
                        (4500, 4500): (3332, 3332), (20, 20): (56, 56), (3360, 3360): (3360, 3360)}.items():
        assert oip.smart_resize(h, w) == exp


def _hf_qwen2(cfg: DotsConfig, sd):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    hc = Qwen2Config(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps,
        rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
        max_position_embeddings=4096, tie_word_embeddings=False, attention_dropout=0.0,
        use_sliding_window=False)
    hc._attn_implementation = "eager"
    m = Qwen2ForCausalLM(hc).eval().float()
    lm_sd = {k: v.float() for k, v in sd.items() if k.startswith("model.") or k.startswith("lm_head.")}
    missing, unexpected = m.load_state_dict(lm_sd, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    return m


def test_lm_matches_transformers_qwen2():
    cfg = DotsConfig.tiny(layers=2, vocab=512)
    sd = random_state_dict(cfg, seed=3)
    hf = _hf_qwen2(cfg, sd)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, cfg.vocab_size - 8, (37,), generator=g)
    with torch.no_grad():
        ref = hf(input_ids=ids[None]).logits[0]
        emb = om.build_embeds(sd, cfg, ids, None)
        got = om.lm_forward(sd, cfg, emb, om.KVCache(cfg.num_hidden_layers), last_only=False)
    assert torch.allclose(got, ref, atol=2e-4, rtol=1e-4), (got - ref).abs().max()
    # greedy loop == GenerationMixin.generate(do_sample=False), incl. the KV-cache decode steps
    with torch.no_grad():
        hf_out = hf.generate(input_ids=ids[None], max_new_tokens=12, do_sample=False,
                             eos_token_id=None, pad_token_id=0)[0, len(ids):].tolist()
    mine = om.generate(sd, cfg, ids, None, None, 12)
    assert mine == hf_out


def test_lm_inputs_embeds_path_matches_transformers():
    # the vision rows enter through inputs_embeds
    cfg = DotsConfig.tiny(layers=1, vocab=512)
    sd = random_state_dict(cfg, seed=5)
    hf = _hf_qwen2(cfg, sd)
    ids = torch.arange(20) % 100
    ids[3:11] = cfg.image_token_id
    vis = torch.randn(8, cfg.hidden_size, generator=torch.Generator().manual_seed(1)) * 0.05
    emb = om.build_embeds(sd, cfg, ids, vis)
    with torch.no_grad():
        ref = hf(inputs_embeds=emb[None]).logits[0, -1]
        got = om.lm_forward(sd, cfg, emb, om.KVCache(1))[0]
    assert torch.allclose(got, ref, atol=2e-4, rtol=1e-4)


def test_image_processor_matches_transformers_pil():
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    from PIL import Image
    rng = np.random.default_rng(0)
    for (w, h) in [(200, 120), (333, 517), (56, 56), (640, 36)]:
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
        proc = Qwen2VLImageProcessorPil(min_pixels=3136, max_pixels=11289600, patch_size=14,
                                        temporal_patch_size=1, merge_size=2)
        ref = proc(images=[img], return_tensors="np")
        pv, thw = oip.preprocess(img)
        assert list(ref["image_grid_thw"][0]) == list(thw)
        assert pv.shape == ref["pixel_values"].shape
        np.testing.assert_allclose(pv, ref["pixel_values"], atol=1e-6, rtol=0)


def test_vision_rope_ids_match_transformers():
    from transformers.vision_utils import get_vision_position_ids
    grid = torch.tensor([[1, 4, 6], [1, 8, 2], [1, 2, 2]])
    assert torch.equal(om.vision_position_ids(grid, 2), get_vision_position_ids(grid, 2))


def test_vision_rotary_matches_transformers():
    from transformers.models.qwen2_vl.modeling_qwen2_vl import VisionRotaryEmbedding, apply_rotary_pos_emb_vision
    from transformers.vision_utils import get_vision_position_ids
    grid = torch.tensor([[1, 4, 6]])
    D = 128
    pos = get_vision_position_ids(grid, 2)
    freqs = VisionRotaryEmbedding(D // 2)(pos)
    emb = torch.cat((freqs, freqs), dim=-1)
    q = torch.randn(24, 3, D)
    k = torch.randn(24, 3, D)
    rq, rk = apply_rotary_pos_emb_vision(q, k, emb.cos(), emb.sin())
    cos, sin = om.vision_rope_cos_sin(grid, D, 2)
    mq = q * cos.unsqueeze(1) + om.rotate_half(q) * sin.unsqueeze(1)
    assert torch.allclose(mq, rq, atol=1e-6)


def test_patch_merger_matches_transformers():
    from transformers.models.qwen2_vl.modeling_qwen2_vl import PatchMerger
    cfg = DotsConfig.tiny()
    v = cfg.vision
    pm = PatchMerger(dim=v.hidden_size, context_dim=v.embed_dim, spatial_merge_size=2).eval()
    sd = {"vision_tower.merger." + k: t for k, t in pm.state_dict().items()}
    x = torch.randn(16, v.embed_dim)
    with torch.no_grad():
        ref = pm(x)
    # run only the merger tail of the oracle tower
    y = om.layer_norm(x, sd["vision_tower.merger.ln_q.weight"], sd["vision_tower.merger.ln_q.bias"], 1e-6, False)
    y = y.view(-1, v.embed_dim * 4)
    y = torch.nn.functional.gelu(om.linear(y, sd["vision_tower.merger.mlp.0.weight"], sd["vision_tower.merger.mlp.0.bias"]))
    y = om.linear(y, sd["vision_tower.merger.mlp.2.weight"], sd["vision_tower.merger.mlp.2.bias"])
    assert torch.allclose(y, ref, atol=1e-5)


def test_vision_tower_runs_and_modes_agree():
    cfg = DotsConfig.tiny()
    sd = random_state_dict(cfg, seed=1)
    grid = torch.tensor([[1, 4, 6], [1, 2, 4]])
    n = int((grid[:, 1] * grid[:, 2]).sum())
    pv = torch.randn(n, cfg.vision.patch_dim, generator=torch.Generator().manual_seed(2))
    a = om.vision_tower(sd, cfg, pv, grid, emulate_bf16=False)
    b = om.vision_tower(sd, cfg, pv, grid, emulate_bf16=True)
    assert a.shape == (n // 4, cfg.hidden_size)
    assert (a - b).abs().max() < 0.1 * a.abs().max()
    # images are independent: the second image alone gives the same rows
    c = om.vision_tower(sd, cfg, pv[24:], grid[1:], emulate_bf16=False)
    assert torch.allclose(c, a[6:], atol=1e-4)


def test_fixed_point_bicubic_restatement_is_bit_exact_with_pillow():
    """oracle.pil_bicubic_resize (the checker of the GPU preprocessing kernels) == Pillow's own BICUBIC resize."""
    from PIL import Image
    rng = np.random.default_rng(3)
    for (w, h, rw, rh) in [(100, 40, 112, 56), (333, 517, 336, 504), (640, 36, 644, 28), (50, 50, 28, 28), (200, 300, 200, 308)]:
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
        ref = np.asarray(img.resize((rw, rh), resample=Image.BICUBIC))
        assert np.array_equal(oip.pil_bicubic_resize(np.asarray(img), rw, rh), ref), (w, h, rw, rh)


def test_vision_block_composition_matches_transformers_qwen2_5_vl_block():
    """One whole transformer block of the oracle's vision tower == the in-container transformers Qwen2_5_VLVisionBlock on
    shared weights: RMSNorm -> qkv(+bias) -> 2-D rope -> per-image bidirectional attention -> proj -> +res -> RMSNorm ->
    SwiGLU(gate, up -> down) -> +res.  This pins the block COMPOSITION (norm order, residual placement, qkv split order,
    rotate_half convention, var-len attention boundaries, SwiGLU wiring), which VERDICT r1 flagged as recollection-only;
    dots.ocr's own block (HF-hub remote code, absent offline) is recalled to differ from it only in bias-free linears and
    eps, both of which oracle.vision_block reads from the state dict / config."""
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLVisionBlock
    from transformers.models.qwen2_vl.modeling_qwen2_vl import VisionRotaryEmbedding
    from transformers.vision_utils import get_vision_position_ids
    torch.manual_seed(3)
    E, Hh, I = 256, 2, 384
    vc = Qwen2_5_VLVisionConfig(hidden_size=E, num_heads=Hh, intermediate_size=I, depth=1, hidden_act="silu", out_hidden_size=E)
    vc._attn_implementation = "sdpa"
    blk = Qwen2_5_VLVisionBlock(vc).eval().float()
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(torch.randn_like(prm) * (0.05 if prm.dim() > 1 else 0.3) + (1.0 if prm.dim() == 1 and prm.shape[0] == E and prm is blk.norm1.weight else 0.0))
    hf = blk.state_dict()
    sd = {"b.norm1.weight": hf["norm1.weight"], "b.norm2.weight": hf["norm2.weight"],
          "b.attn.qkv.weight": hf["attn.qkv.weight"], "b.attn.qkv.bias": hf["attn.qkv.bias"],
          "b.attn.proj.weight": hf["attn.proj.weight"], "b.attn.proj.bias": hf["attn.proj.bias"],
          "b.mlp.fc1.weight": hf["mlp.gate_proj.weight"], "b.mlp.fc1.bias": hf["mlp.gate_proj.bias"],
          "b.mlp.fc3.weight": hf["mlp.up_proj.weight"], "b.mlp.fc3.bias": hf["mlp.up_proj.bias"],
          "b.mlp.fc2.weight": hf["mlp.down_proj.weight"], "b.mlp.fc2.bias": hf["mlp.down_proj.bias"]}
    grid = torch.tensor([[1, 4, 6], [1, 2, 8], [1, 2, 2]])                     # three images of 24, 16 and 4 patches
    lens = (grid[:, 1] * grid[:, 2]).tolist()
    n = sum(lens)
    x = torch.randn(n, E)
    D = E // Hh
    freqs = VisionRotaryEmbedding(D // 2)(get_vision_position_ids(grid, 2))
    emb = torch.cat((freqs, freqs), dim=-1)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    with torch.no_grad():
        ref = blk(x, cu_seqlens=cu, position_embeddings=(emb.cos(), emb.sin()))
    cos, sin = om.vision_rope_cos_sin(grid, D, 2)
    got = om.vision_block(sd, "b.", x, cos.unsqueeze(1), sin.unsqueeze(1), lens, Hh, D, 1e-6, False)
    assert torch.allclose(got, ref, atol=2e-5), (got - ref).abs().max()
    # attention really is per image: moving the image boundaries changes the result
    other = om.vision_block(sd, "b.", x, cos.unsqueeze(1), sin.unsqueeze(1), [n], Hh, D, 1e-6, False)
    assert (other - ref).abs().max() > 1e-3


def test_fp8_oracle_mode_definitions():
    """The fp8 oracle mode's building blocks (what tests/test_fp8_gpu.py holds the engine to): per-row e4m3 quantisation is exact on
    representable rows, saturates nowhere, leaves zero rows alone; a8 linear == explicit quantise-dequantise-matmul; the state-dict
    transform touches exactly the linears (and copies a tied lm_head) and is idempotent."""
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.weights import random_state_dict
    g = torch.Generator().manual_seed(3)
    w = torch.randn(37, 96, generator=g) * torch.logspace(-4, 2, 37).view(-1, 1)
    w[5] = 0
    q, s = om.quantize_rows_fp8(w)
    assert q.abs().max() <= 448 and torch.isfinite(q).all()
    assert torch.equal(q[5], torch.zeros(96)) and float(s[5]) == 1.0
    assert torch.allclose(q.abs().amax(1)[s != 1.0], torch.full((36,), 448.0))            # the row maximum maps to the format's maximum
    rel = ((q * s[:, None] - w).abs() / w.abs().amax(1, keepdim=True).clamp_min(1e-30)).max()
    assert float(rel) <= 2 ** -4 + 1e-6                                                     # half an ulp of a 3-bit mantissa, relative to the row max
    q2, s2 = om.quantize_rows_fp8(q * s[:, None])                                           # representable rows survive unchanged
    assert torch.equal(q2, q) and torch.allclose(s2, s)
    x = torch.randn(11, 96, generator=g)
    xq, xs = om.quantize_rows_fp8(x)
    assert torch.equal(om.linear(x, w, None, a8=True), (xq * xs[:, None]) @ w.t())
    cfg = DotsConfig.tiny(layers=1, v_layers=1)
    sd = random_state_dict(cfg, seed=1)
    qsd = om.quantize_fp8_state_dict(sd)
    changed = sorted(k for k in sd if not torch.equal(qsd[k].float(), sd[k].float()))
    assert changed and all(k.endswith(om.FP8_LINEAR_SUFFIXES) for k in changed)
    assert not any(k.startswith("vision_tower.patch_embed") or "norm" in k or k.endswith(".bias") or k == "model.embed_tokens.weight" for k in changed)
    again = om.quantize_fp8_state_dict(qsd)
    assert all(torch.allclose(again[k].float(), qsd[k].float(), rtol=1e-6, atol=0) for k in qsd)
    tied = {k: v for k, v in sd.items() if k != "lm_head.weight"}
    assert "lm_head.weight" in om.quantize_fp8_state_dict(tied) and torch.equal(tied["model.embed_tokens.weight"], sd["model.embed_tokens.weight"])
