"""The two launch plans of the decode step must produce EXACTLY the same bits: the whole-chip plan (qkv / o_proj / down_proj as 8-row half
tiles, one gate|up workgroup per tile pair) and the PARTITION plan (whole 16-row tiles; gate|up as one resident round of workgroups that
walk the tile pairs) that dots_generate / dots_slots_decode replay on the decode CU partition beside a prefetched vision tower — same
arithmetic per element, only the work decomposition differs.  Compared through the C ABI (dots_set_decode_plan): fp32 logits of every
step, bitwise.  (Rounds 3-4 also carried fused decode launches with in-launch hand-offs, experiments/decode_flow/: bit-identical,
break-even, removed.)

Cases: contexts that put the step's token at the last key of a page, the first key of a new page and in the middle; sequences of
different lengths in one batch (idle KV splits); contexts beyond 4 x 64 pages (a wave walks several pages); fp8 weights; graph replay;
9 and 20 rows (16-row X images, two batch tiles); the real dimensions, where the partition holds fewer gate|up workgroups (3 per CU x
128 CUs) than there are tile pairs (560), so the walking path really runs.
"""
import numpy as np
import pytest
import torch

from dots_ocr_amd.config import DotsConfig
from dots_ocr_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu


def _prompts(cfg, lens, seed):
    rng = np.random.default_rng(seed)
    hi = min(cfg.vocab_size, cfg.image_token_id) - 1
    ids = [rng.integers(0, hi, n).astype(np.int32) for n in lens]
    return np.concatenate(ids), np.asarray(lens, np.int32)


def _decode(eng, mode, ids, lens, n_steps):
    eng.set_decode_plan(mode)
    eng.prefill(ids, lens)
    logits, tokens = [eng.get_logits().copy()], [eng.get_last_tokens().copy()]
    for _ in range(n_steps):
        eng.decode_step()
        logits.append(eng.get_logits().copy())
        tokens.append(eng.get_last_tokens().copy())
    return logits, tokens


def _assert_same(ref, got, what):
    for s, (a, b) in enumerate(zip(ref[0], got[0])):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{what}: logits differ at step {s} (max |d| {np.abs(a - b).max():.3e})"
    for s, (a, b) in enumerate(zip(ref[1], got[1])):
        assert np.array_equal(a, b), f"{what}: tokens differ at step {s}"


@pytest.fixture(scope="module")
def tiny():
    from dots_ocr_amd.engine import Engine
    cfg = DotsConfig.tiny(layers=3, v_layers=2, vocab=1024)
    sd = random_state_dict(cfg, seed=5)
    eng = Engine(cfg, max_batch=9, max_seq_len=1024, max_patches=1024, max_prefill_tokens=4096)
    eng.load_state_dict(sd)
    yield cfg, sd, eng
    eng.close()


@pytest.mark.parametrize("lens", [[70], [63, 64, 65, 1, 127, 128, 200, 190], [5, 300, 61]])
def test_partition_plan_equals_whole_chip_plan_bitwise(tiny, lens):
    """6 steps so that several sequences cross a page boundary while decoding."""
    cfg, sd, eng = tiny
    ids, ln = _prompts(cfg, lens, seed=len(lens))
    ref = _decode(eng, 0, ids, ln, 6)
    _assert_same(ref, _decode(eng, 1, ids, ln, 6), f"partition plan, prompt lengths {lens}")


def test_plans_under_graph_replay_and_generate(tiny):
    """dots_generate replays ONE captured graph per plan."""
    cfg, sd, eng = tiny
    ids, ln = _prompts(cfg, [40, 90, 64, 33], seed=9)
    out = {}
    for mode in (0, 1):
        eng.set_decode_plan(mode)
        out[mode] = eng.generate(ids, ln, max_new_tokens=80)
    eng.set_decode_plan(0)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("lens", [[63, 64, 65, 1, 127, 128, 200, 190], [5, 300, 61]])
def test_streaming_attention_plans_bitwise(tiny, lens):
    """Round 5: + 2 forces the streaming decode-attention kernel, + 4 the per-split one; logits of every step bit for bit, on the whole-chip
    plan (4 vs 2) and on the partition plan (5 vs 3), eager steps and graph replay."""
    cfg, sd, eng = tiny
    ids, ln = _prompts(cfg, lens, seed=len(lens))
    try:
        ref = _decode(eng, 4, ids, ln, 6)
        _assert_same(ref, _decode(eng, 2, ids, ln, 6), f"streaming attention, prompt lengths {lens}")
        _assert_same(ref, _decode(eng, 3, ids, ln, 6), f"streaming attention on the partition plan, prompt lengths {lens}")
        _assert_same(ref, _decode(eng, 5, ids, ln, 6), f"per-split attention on the partition plan, prompt lengths {lens}")
        eng.set_decode_plan(4)
        a = eng.generate(ids, ln, max_new_tokens=70)
        eng.set_decode_plan(2)
        b = eng.generate(ids, ln, max_new_tokens=70)
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    finally:
        eng.set_decode_plan(0)


def test_batch_of_nine(tiny):
    cfg, sd, eng = tiny
    ids, ln = _prompts(cfg, [20 + 3 * i for i in range(9)], seed=2)
    ref = _decode(eng, 0, ids, ln, 3)
    _assert_same(ref, _decode(eng, 1, ids, ln, 3), "B = 9, partition plan (16-row X image)")


def test_partition_plan_above_16_rows_bitwise():
    """Round 4: the partition plan (whole-tile projections, pair-walking gate|up) is no longer limited to 8 rows — continuous batching
    with 9-64 occupied slots replays it beside a prefetched tower.  20 rows = two 16-row batch tiles."""
    from dots_ocr_amd.engine import Engine
    cfg = DotsConfig.tiny(layers=2, v_layers=2, vocab=1024)
    sd = random_state_dict(cfg, seed=12)
    eng = Engine(cfg, max_batch=20, max_seq_len=512, max_patches=256, max_prefill_tokens=4096)
    eng.load_state_dict(sd)
    ids, ln = _prompts(cfg, [17 + 11 * i for i in range(20)], seed=13)
    ref = _decode(eng, 0, ids, ln, 4)
    _assert_same(ref, _decode(eng, 1, ids, ln, 4), "B = 20, partition plan")
    eng.close()


def test_plans_walk_several_pages_per_wave_beyond_16k_context():
    """max_seq_len > 16 384 caps the KV split at 64 workgroups of 4 waves: a wave then walks pages p, p + 256, ..."""
    from dots_ocr_amd.engine import Engine
    cfg = DotsConfig.tiny(layers=2, v_layers=2, vocab=1024)
    sd = random_state_dict(cfg, seed=6)
    eng = Engine(cfg, max_batch=2, max_seq_len=17408, max_patches=256, max_prefill_tokens=33000)
    eng.load_state_dict(sd)
    ids, ln = _prompts(cfg, [16500, 16383], seed=4)
    ref = _decode(eng, 0, ids, ln, 3)
    _assert_same(ref, _decode(eng, 1, ids, ln, 3), "16.5k-token contexts, partition plan")
    eng.close()


def test_partition_plan_fp8_bitwise():
    from dots_ocr_amd.engine import Engine
    cfg = DotsConfig.tiny(layers=2, v_layers=2, vocab=1024)
    sd = random_state_dict(cfg, seed=7)
    eng = Engine(cfg, max_batch=4, max_seq_len=512, max_patches=256, fp8_weights=True)
    eng.load_state_dict(sd)
    ids, ln = _prompts(cfg, [64, 100, 31], seed=8)
    ref = _decode(eng, 0, ids, ln, 5)
    ref2 = _decode(eng, 0, ids, ln, 5)
    _assert_same(ref, ref2, "fp8, whole-chip plan run twice")
    _assert_same(ref, _decode(eng, 1, ids, ln, 5), "fp8, partition plan")
    eng.close()


def test_plans_at_the_real_dimensions_bitwise():
    """dots.ocr's LM dimensions (H 1536, 12:2 heads, I 8960, vocab 151 936; 4 layers of seeded random weights cover every role at its
    real shape): B = 8, one context of 5.2 k tokens (25 KV splits like the bench) beside short ones that sit on, before and after a
    page boundary; 5 steps."""
    from dots_ocr_amd.engine import Engine
    cfg = DotsConfig()
    cfg.num_hidden_layers = 4
    cfg.vision.num_hidden_layers = 1
    sd = random_state_dict(cfg, seed=1, threads=16)
    lens = [5247, 64, 1, 700, 63, 65, 128, 129]
    eng = Engine(cfg, max_batch=8, max_seq_len=6288, max_patches=256, max_prefill_tokens=sum(lens) + 64)
    eng.load_state_dict(sd)
    ids, ln = _prompts(cfg, lens, seed=3)
    ref = _decode(eng, 0, ids, ln, 5)
    _assert_same(ref, _decode(eng, 1, ids, ln, 5), "real dimensions, partition plan")
    _assert_same(ref, _decode(eng, 3, ids, ln, 5), "real dimensions, partition plan with the streaming attention kernel (round 5)")
    _assert_same(ref, _decode(eng, 6 - 2, ids, ln, 5), "real dimensions, per-split attention forced")
    eng.close()
