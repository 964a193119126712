"""Whole-tower ViT parity at the REAL dots.ocr dimensions, inside the driver-run GPU suite (VERDICT r2 missing #3 / next #2).

The hub `DotsVisionTransformer.forward` the reference loads at dots_ocr/parser.py:68-74 is 42 blocks x 1536 wide x 12 heads x
4224 MLP; every other whole-tower test in this suite runs DotsConfig.tiny (256 wide, 2 heads, 3 blocks).  Here the engine's
`dots_vit_forward` is compared with oracle/model.py `vision_tower` — the ORACLE RUNNING ITS OWN TOWER on the same pixels — for
  * a 583x550 page (1 680 patches; the A4 size — 19 824 patches — is held to a committed oracle run by test_a4_anchor_gpu.py): merged embeddings within 3 % of the tensor's max
    magnitude vs the bf16-emulated oracle and 6 % vs the fp32 oracle (the tolerances of DESIGN §2 / tests/test_model_gpu.py);
  * the per-block residual-stream error (dots_debug_capture_hidden vs the oracle's return_hidden) written to the report, with
    the assertion that no single block adds more than 1 % of the stream's RMS vs the emulated oracle (a wrong kernel at 12
    heads / E = 1536 / the 6144-wide merger shows up as a jump, bf16 noise as a smooth walk);
  * 16 teacher-forced greedy decode steps after prefill on the 583x550 page, the oracle computing vision rows, prefill and
    decode itself: max |logit error| vs the fp32 oracle <= 0.125, tokens equal wherever the emulated oracle's margin > 0.25;
  * the same page through an `fp8_weights` engine vs the fp8 oracle (quantised state dict, per-token e4m3 activations) AND vs
    the unquantised fp32 oracle: stated tolerances below, the measured numbers go to the report.
Report: gpurun_out/r04_vit_fullwidth_parity.json (copied to profiles/ by hand).
"""
import json
import os
import time
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import model as om

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
REPORT = {}


def _bf(u16):
    return torch.from_numpy((u16.astype(np.uint32) << 16).view(np.float32))


@pytest.fixture(scope="module")
def world():
    from dots_ocr_amd.config import DotsConfig
    from shared_weights import F32View, full_sd
    cfg = DotsConfig()
    threads = min(os.cpu_count() or 8, 64)
    torch.set_num_threads(threads)
    sd = full_sd(0)                              # shared with the other full-size test modules of the session (never modified in place)
    sd32 = F32View(sd)                           # fp32 on first use, kept for the session
    yield cfg, sd, sd32
    out = ROOT / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "r04_vit_fullwidth_parity.json").write_text(json.dumps(REPORT, indent=1))
    except OSError:
        pass


def _page(index, size):
    from dots_ocr_amd.image_utils import preprocess_image
    from dots_ocr_amd.synthetic import synth_page
    pv, thw = preprocess_image(synth_page(index, size))
    return pv, thw


# (round 6: the 946x1024 / 5 032-patch case — 125 s of inline oracle time — is replaced by tests/test_a4_anchor_gpu.py, which holds the whole tower at
# 19 824 patches to a committed full-depth oracle run; the per-block trace stays here at the size the oracle finishes in seconds)
@pytest.mark.parametrize("size,patches", [((583, 550), 1680)])
def test_whole_tower_at_real_width_matches_the_oracles_own_tower(world, size, patches):
    from dots_ocr_amd.engine import Engine
    cfg, sd, sd32 = world
    v = cfg.vision
    assert (v.num_hidden_layers, v.embed_dim, v.num_attention_heads, v.intermediate_size) == (42, 1536, 12, 4224)
    pv, thw = _page(3, size)
    N = pv.shape[0]
    assert N == patches
    eng = Engine(cfg, max_batch=1, max_seq_len=2048, max_patches=N + 64, max_prefill_tokens=2048)
    eng.load_state_dict(sd)
    eng.capture_hidden(v.num_hidden_layers * N * v.embed_dim)
    out = torch.empty(N // 4, cfg.hidden_size, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    eng.vit_forward(pv, np.asarray([thw], np.int64), out_dev=out.data_ptr())
    eng.synchronize()
    got = out.float().cpu()
    blocks = [_bf(eng.read_hidden("vit", i)) for i in range(v.num_hidden_layers)]
    eng.close()

    t0 = time.perf_counter()
    emu, emu_h = om.vision_tower(sd32, cfg, torch.from_numpy(pv), torch.tensor([thw]), emulate_bf16=True, return_hidden=True)
    t1 = time.perf_counter()
    f32, f32_h = om.vision_tower(sd32, cfg, torch.from_numpy(pv), torch.tensor([thw]), emulate_bf16=False, return_hidden=True)
    t2 = time.perf_counter()
    scale = f32.abs().max().item()
    e_emu = (got - emu).abs().max().item() / scale
    e_f32 = (got - f32).abs().max().item() / scale
    trace, prev, worst_jump = [], 0.0, 0.0
    for i, g in enumerate(blocks):
        rms = float(f32_h[i].pow(2).mean().sqrt())
        r_emu = float((g - emu_h[i]).pow(2).mean().sqrt()) / rms
        r_f32 = float((g - f32_h[i]).pow(2).mean().sqrt()) / rms
        trace.append({"block": i, "rms_err_vs_emulated": r_emu, "rms_err_vs_fp32": r_f32,
                      "max_err_vs_emulated": float((g - emu_h[i]).abs().max()) / rms})
        worst_jump = max(worst_jump, r_emu - prev)
        prev = r_emu
    REPORT[f"tower_{size[0]}x{size[1]}"] = {
        "patches": N, "merged_rows": N // 4, "max_err_over_max_magnitude": {"vs_emulated_bf16": e_emu, "vs_fp32": e_f32},
        "tolerance": "3 % (emulated) / 6 % (fp32) of the output's max magnitude; no block adds > 1 % RMS vs the emulated oracle",
        "largest_single_block_rms_increase_vs_emulated": worst_jump, "oracle_seconds": {"emulated": t1 - t0, "fp32": t2 - t1},
        "per_block": trace}
    print(f"{size}: merged rel err {e_emu:.4f} (emulated) / {e_f32:.4f} (fp32); residual-stream RMS err after block 41: "
          f"{trace[-1]['rms_err_vs_emulated']:.4f} / {trace[-1]['rms_err_vs_fp32']:.4f}; largest single-block increase {worst_jump:.4f}")
    assert e_emu < 0.03 and e_f32 < 0.06
    assert worst_jump < 0.01, f"one block adds {worst_jump:.4f} of the stream's RMS vs the emulated oracle"


def _run_engine(cfg, sd, pv, thw, ids, n_steps, fp8):
    from dots_ocr_amd.engine import Engine
    eng = Engine(cfg, max_batch=1, max_seq_len=len(ids) + 128, max_patches=pv.shape[0] + 64, max_prefill_tokens=len(ids) + 64, fp8_weights=fp8)
    eng.load_state_dict(sd)
    eng.vit_forward(pv, np.asarray([thw], np.int64))
    eng.prefill(ids, np.asarray([len(ids)], np.int32))
    lg, tk = [eng.get_logits()[0].copy()], [int(eng.get_last_tokens()[0])]
    for _ in range(1, n_steps):
        eng.decode_step()
        lg.append(eng.get_logits()[0].copy())
        tk.append(int(eng.get_last_tokens()[0]))
    eng.close()
    return lg, tk


def test_page_through_tower_prefill_and_decode_with_the_oracle_running_its_own_tower(world):
    """bf16 engine: pixels -> tower -> prefill -> 16 greedy steps; the oracle does the same from the same pixels (its own
    vision rows, NOT the engine's), teacher-forced on the engine's tokens."""
    from dots_ocr_amd.synthetic import synth_prompt_ids
    cfg, sd, sd32 = world
    n_steps = 16
    pv, thw = _page(3, (583, 550))
    ids = synth_prompt_ids(cfg, pv.shape[0] // 4, n_text_tokens=64, seed=3)
    lg, tk = _run_engine(cfg, sd, pv, thw, ids, n_steps, fp8=False)
    t_ids, t_pv, t_thw = torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(pv), torch.tensor([thw])
    _, emu = om.generate(sd32, cfg, t_ids, t_pv, t_thw, n_steps, emulate_bf16=True, forced_tokens=tk, return_logits=True)
    _, f32 = om.generate(sd32, cfg, t_ids, t_pv, t_thw, n_steps, emulate_bf16=False, forced_tokens=tk, return_logits=True)
    worst, agree, checked, rows = 0.0, 0, 0, []
    for s in range(n_steps):
        e = torch.from_numpy(lg[s]).double()
        d32 = float((e - f32[s].double()).abs().max())
        top2 = torch.topk(emu[s], 2)
        margin = float(top2.values[0] - top2.values[1])
        same = tk[s] == int(top2.indices[0])
        agree += int(same)
        worst = max(worst, d32)
        rows.append({"step": s, "max_abs_err_vs_fp32": d32, "max_abs_err_vs_emulated": float((e - emu[s].double()).abs().max()),
                     "oracle_top2_margin": margin, "token_equal": bool(same)})
        if margin > 0.25:
            checked += 1
            assert same, f"step {s}: engine token {tk[s]} != oracle {int(top2.indices[0])} at margin {margin:.3f}"
    REPORT["bf16_page_583x550_end_to_end"] = {
        "steps": n_steps, "tokens_equal_to_emulated_oracle_argmax": agree, "steps_with_margin_above_0.25": checked,
        "max_abs_logit_err_vs_fp32_oracle": worst, "logit_range": float(f32[0].max() - f32[0].min()),
        "tolerance": "max |logit err| vs fp32 oracle <= 0.125; tokens equal where the emulated oracle's margin > 0.25", "per_step": rows}
    print(f"bf16 end to end (oracle tower): {agree}/{n_steps} tokens equal, max |logit err| vs fp32 {worst:.4f}")
    assert worst <= 0.125
    assert agree >= n_steps - 2


def test_fp8_engine_at_real_dimensions_vs_fp8_oracle_and_vs_the_unquantised_oracle(world):
    """fp8 configuration (BASELINE configs[4]) at 42 / 28 layers.  Three distances, all stated:
      * engine vs the fp8 oracle (same quantised weights, per-token e4m3 activations in tower + prefill): the kernels' error.
        Activation quantisation is a STEP function: a bf16-sized difference in an activation flips some e4m3 roundings (each flip
        moves that element by 2^-3 relative), and 70 layers amplify it.  The yardstick is therefore the fp8 oracle's OWN sensitivity
        to bf16 rounding — the distance between its bf16-emulated and its fp32 evaluation of the same quantised model (measured
        here: both oracles run): the engine must be no farther from either than 1.5 x that distance (floor 4 % of the logit range).
      * engine vs the UNQUANTISED fp32 oracle: what fp8 costs the model — not a kernel property; asserted only loosely
        (<= 25 % of the logit range on these random weights, whose logits are nearly flat) and REPORTED."""
    from dots_ocr_amd.synthetic import synth_prompt_ids
    cfg, sd, sd32 = world
    n_steps = 8
    pv, thw = _page(3, (583, 550))
    ids = synth_prompt_ids(cfg, pv.shape[0] // 4, n_text_tokens=64, seed=3)
    lg, tk = _run_engine(cfg, sd, pv, thw, ids, n_steps, fp8=True)
    t_ids, t_pv, t_thw = torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(pv), torch.tensor([thw])
    qsd = om.quantize_fp8_state_dict(sd32)
    _, q_emu = om.generate(qsd, cfg, t_ids, t_pv, t_thw, n_steps, emulate_bf16=True, forced_tokens=tk, return_logits=True, fp8_act=True)
    _, q_f32 = om.generate(qsd, cfg, t_ids, t_pv, t_thw, n_steps, emulate_bf16=False, forced_tokens=tk, return_logits=True, fp8_act=True)
    del qsd
    _, u_f32 = om.generate(sd32, cfg, t_ids, t_pv, t_thw, n_steps, emulate_bf16=False, forced_tokens=tk, return_logits=True)
    w_emu = w_f32 = w_unq = w_oo = 0.0
    same_unq = 0
    for s in range(n_steps):
        got = torch.from_numpy(lg[s])
        rng = float(u_f32[s].max() - u_f32[s].min())
        w_emu = max(w_emu, float((got - q_emu[s]).abs().max()) / rng)
        w_f32 = max(w_f32, float((got - q_f32[s]).abs().max()) / rng)
        w_unq = max(w_unq, float((got - u_f32[s]).abs().max()) / rng)
        w_oo = max(w_oo, float((q_emu[s] - q_f32[s]).abs().max()) / rng)
        same_unq += int(tk[s] == int(u_f32[s].argmax()))
    REPORT["fp8_page_583x550_end_to_end"] = {
        "steps": n_steps, "max_err_over_logit_range": {"vs_fp8_oracle_emulated": w_emu, "vs_fp8_oracle_fp32": w_f32, "vs_unquantised_fp32_oracle": w_unq,
                                              "fp8_oracle_emulated_vs_fp8_oracle_fp32": w_oo},
        "tokens_equal_to_unquantised_oracle_argmax": same_unq,
        "tolerance": "vs either fp8 oracle: max(4 %, 1.5 x the distance between the two fp8 oracles) of the logit range; vs the unquantised oracle "
                     "reported, loosely bounded at 25 %"}
    print(f"fp8 at real dims: {w_emu:.4f} / {w_f32:.4f} of the logit range vs the fp8 oracle (emulated / fp32; the two oracles differ by {w_oo:.4f}), "
          f"{w_unq:.4f} vs the unquantised fp32 oracle; "
          f"{same_unq}/{n_steps} tokens equal the unquantised oracle's arg max")
    tol = max(0.04, 1.5 * w_oo)
    assert w_emu < tol and w_f32 < tol, f"engine {w_emu:.4f} / {w_f32:.4f} vs tolerance {tol:.4f}"
    assert w_unq < 0.25


def test_fp8_engine_layer_by_layer_with_resynchronisation(world):
    """VERDICT r3 #4b: the end-to-end fp8 comparison above passes by being inside the oracle's own noise (per-token e4m3 activation
    quantisation is a step function; 70 layers amplify every flipped rounding), so it bounds nothing a kernel bug below ~14 % of the
    logit range could do.  Here every layer is compared ON ITS OWN: the fp8 oracle computes vision block i / LM layer i from the
    ENGINE's captured residual stream after block / layer i-1 (dots_debug_capture_hidden), so a flipped rounding cannot travel further
    than the layer it happens in.  Tolerance back to the bf16 one: <= 3 % of the layer output's max magnitude, every layer."""
    import copy
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.synthetic import synth_prompt_ids
    cfg, sd, sd32 = world
    v = cfg.vision
    pv, thw = _page(3, (583, 550))
    N = pv.shape[0]
    ids = synth_prompt_ids(cfg, N // 4, n_text_tokens=64, seed=3)
    T = len(ids)
    eng = Engine(cfg, max_batch=1, max_seq_len=T + 128, max_patches=N + 64, max_prefill_tokens=T + 64, fp8_weights=True)
    eng.load_state_dict(sd)
    eng.capture_hidden(v.num_hidden_layers * N * v.embed_dim + cfg.num_hidden_layers * T * cfg.hidden_size)
    eng.vit_forward(pv, np.asarray([thw], np.int64))
    eng.prefill(ids, np.asarray([T], np.int32))
    eng.synchronize()
    vit = [_bf(eng.read_hidden("vit", i)) for i in range(v.num_hidden_layers)]
    lm = [_bf(eng.read_hidden("lm", i)) for i in range(cfg.num_hidden_layers)]
    eng.close()
    assert vit[0].shape == (N, v.embed_dim) and lm[0].shape == (T, cfg.hidden_size)

    qsd = om.quantize_fp8_state_dict(sd32)
    grid = torch.tensor([thw])
    cos, sin = om.vision_rope_cos_sin(grid, v.head_dim, v.spatial_merge_size)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    rows_v, worst_v = [], 0.0
    with torch.no_grad():
        for i in range(1, v.num_hidden_layers):
            ref = om.vision_block(qsd, f"vision_tower.blocks.{i}.", vit[i - 1], cos, sin, [N], v.num_attention_heads, v.head_dim, v.rms_norm_eps, True, a8=True)
            err = float((vit[i] - ref).abs().max()) / float(ref.abs().max())
            rows_v.append({"block": i, "max_err_over_max_magnitude": err})
            worst_v = max(worst_v, err)
        # LM prefill layers: a one-layer model whose layer 0 is layer i, fed the engine's stream after layer i-1
        c1 = copy.deepcopy(cfg)
        c1.num_hidden_layers = 1
        rows_l, worst_l = [], 0.0
        for i in range(1, cfg.num_hidden_layers):
            sub = {k.replace(f"model.layers.{i}.", "model.layers.0."): t for k, t in qsd.items() if k.startswith(f"model.layers.{i}.")}
            sub["model.norm.weight"] = qsd["model.norm.weight"]
            sub["lm_head.weight"] = qsd["lm_head.weight"][:16]                       # the logits of this call are not used
            _, hid = om.lm_forward(sub, c1, lm[i - 1], om.KVCache(1), emulate_bf16=True, return_hidden=True, a8=True)
            err = float((lm[i] - hid[0]).abs().max()) / float(hid[0].abs().max())
            rows_l.append({"layer": i, "max_err_over_max_magnitude": err})
            worst_l = max(worst_l, err)
    REPORT["fp8_layer_by_layer_resynchronised"] = {
        "page": "583x550 (1 680 patches), prompt %d tokens" % T, "worst_vision_block": worst_v, "worst_lm_prefill_layer": worst_l,
        "tolerance": "3 % of the layer output's max magnitude, every vision block 1-41 and every LM prefill layer 1-27, each computed by the fp8 oracle "
                     "(bf16-emulated, per-token e4m3 activations) from the engine's own stream of the layer before",
        "vision_blocks": rows_v, "lm_layers": rows_l}
    print(f"fp8 layer by layer, re-synchronised: worst vision block {worst_v:.4f}, worst LM prefill layer {worst_l:.4f} of the layer's max magnitude")
    assert worst_v < 0.03, f"vision block error {worst_v:.4f} >= 3 %: {max(rows_v, key=lambda r: r['max_err_over_max_magnitude'])}"
    assert worst_l < 0.03, f"LM prefill layer error {worst_l:.4f} >= 3 %: {max(rows_l, key=lambda r: r['max_err_over_max_magnitude'])}"
