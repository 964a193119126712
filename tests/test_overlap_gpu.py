"""Software pipelining across page batches (include/dots_ocr_hip.h "Software pipelining"): the vision tower of batch k+1 on the CU-masked
side stream while batch k decodes on the complementary partition must give EXACTLY the tokens of the sequential calls — same kernels,
same arithmetic, only streams and CU masks differ.  Also: the prefetch must not be overwritten before it is taken, a synchronous
dots_vit_forward may follow a prefetch, and slot prefill consumes taken rows."""
import numpy as np
import pytest
import torch

from dots_ocr_amd.config import DotsConfig
from dots_ocr_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu


def _batch(cfg, grids, seed):
    g = torch.Generator().manual_seed(seed)
    grid = np.asarray(grids, np.int64)
    n = int((grid[:, 1] * grid[:, 2]).sum())
    pv = torch.randn(n, cfg.vision.patch_dim, generator=g).numpy()
    seqs = [np.concatenate([torch.randint(0, cfg.vocab_size - 8, (3,), generator=g).numpy(), np.full((h * w // 4,), cfg.image_token_id),
                            torch.randint(0, cfg.vocab_size - 8, (5,), generator=g).numpy()]).astype(np.int32) for (_, h, w) in grids]
    return pv, grid, np.concatenate(seqs), np.asarray([len(s) for s in seqs], np.int32)


@pytest.fixture(scope="module")
def world():
    from dots_ocr_amd.engine import Engine
    cfg = DotsConfig.tiny(layers=3, v_layers=3, vocab=1024)
    sd = random_state_dict(cfg, seed=13)
    eng = Engine(cfg, max_batch=4, max_seq_len=1024, max_patches=8192, max_prefill_tokens=4096)
    eng.load_state_dict(sd)
    yield cfg, eng
    eng.close()


def test_pipelined_batches_equal_sequential_bitwise(world):
    cfg, eng = world
    batches = [_batch(cfg, gr, 50 + i) for i, gr in enumerate([
        [(1, 24, 32), (1, 16, 16), (1, 32, 40)], [(1, 40, 48), (1, 8, 8)], [(1, 16, 24), (1, 24, 24), (1, 8, 12), (1, 32, 32)], [(1, 48, 48)]])]
    NEW = 200
    seq = [eng.generate(ids, lens, pv, grid, max_new_tokens=NEW) for pv, grid, ids, lens in batches]
    # pixels on the device, as in the bench (the buffers stay alive until the rows are taken)
    dev = [torch.from_numpy(pv).cuda() for pv, _, _, _ in batches]
    torch.cuda.synchronize()
    got = []
    eng.vit_prefetch(dev[0].data_ptr(), batches[0][1], on_device=True)
    for k, (pv, grid, ids, lens) in enumerate(batches):
        eng.vit_take()
        if k + 1 < len(batches):
            eng.vit_prefetch(dev[k + 1].data_ptr(), batches[k + 1][1], on_device=True, after_prefill=bool(k & 1))
        got.append(eng.generate(ids, lens, max_new_tokens=NEW, vision_taken=True))
    for (a, an), (b, bn) in zip(seq, got):
        assert np.array_equal(an, bn) and np.array_equal(a, b)
    # host pixels + EOS polling path
    eos = [int(seq[1][0][0, 40])]
    ref = eng.generate(batches[1][2], batches[1][3], batches[1][0], batches[1][1], max_new_tokens=NEW, eos_ids=eos)
    eng.vit_prefetch(batches[1][0], batches[1][1])
    eng.vit_take()
    eng.vit_prefetch(batches[2][0], batches[2][1], after_prefill=True)          # host pixels, launched behind the prefill below
    out = eng.generate(batches[1][2], batches[1][3], max_new_tokens=NEW, eos_ids=eos, vision_taken=True)
    assert np.array_equal(ref[1], out[1]) and np.array_equal(ref[0], out[0]) and out[1][0] <= 41
    eng.vit_take()
    out2 = eng.generate(batches[2][2], batches[2][3], max_new_tokens=NEW, vision_taken=True)
    assert np.array_equal(out2[0], seq[2][0])


def test_prefetch_protocol_errors_and_mixing_with_the_synchronous_tower(world):
    from dots_ocr_amd.engine import DotsEngineError
    cfg, eng = world
    pv, grid, ids, lens = _batch(cfg, [(1, 16, 16), (1, 8, 8)], 77)
    ref = eng.generate(ids, lens, pv, grid, max_new_tokens=30)
    with pytest.raises(DotsEngineError, match="no prefetched"):
        eng.vit_take()
    eng.vit_prefetch(pv, grid)
    with pytest.raises(DotsEngineError, match="waiting"):
        eng.vit_prefetch(pv, grid)
    # a synchronous tower pass right behind a prefetch (shared workspaces: it must wait for it), then the prefetched rows still intact
    pv2, grid2, ids2, lens2 = _batch(cfg, [(1, 24, 24)], 78)
    ref2 = eng.generate(ids2, lens2, pv2, grid2, max_new_tokens=30)
    eng.vit_take()
    out = eng.generate(ids, lens, max_new_tokens=30, vision_taken=True)
    assert np.array_equal(out[0], ref[0])
    again = eng.generate(ids2, lens2, pv2, grid2, max_new_tokens=30)
    assert np.array_equal(again[0], ref2[0])
    # taken rows feed a slot prefill as well
    from dots_ocr_amd.scheduler import ContinuousBatcher
    eng.vit_prefetch(pv, grid)
    eng.vit_take()
    eng.slots_reset()
    eng.set_eos([])
    eng.slots_prefill([0, 1], ids, lens.tolist(), [30, 30])
    for _ in range(2):
        eng.slots_decode(16)
    fin, n = eng.slots_poll()
    assert fin[0] == 1 and fin[1] == 1
    assert eng.slot_read(0, 30).tolist() == ref[0][0].tolist() and eng.slot_read(1, 30).tolist() == ref[0][1].tolist()
    eng.slot_release(0); eng.slot_release(1)
    # an abandoned prefetch does not wedge the engine: slots_reset drops it
    eng.vit_prefetch(pv, grid)
    eng.slots_reset()
    eng.vit_prefetch(pv, grid)
    eng.vit_take()
    assert np.array_equal(eng.generate(ids, lens, max_new_tokens=30, vision_taken=True)[0], ref[0])


def test_scheduler_look_ahead_gives_the_same_tokens(world):
    """ContinuousBatcher(prefetch = k): the towers of the next k queued requests run on the side stream beside the occupied slots (which then
    decode on their CU partition, half-chip plan), the group is admitted with take + prefill only — same tokens as without look-ahead."""
    from dots_ocr_amd.scheduler import ContinuousBatcher, Request
    cfg, eng = world
    grids = [(1, 24, 32), (1, 8, 8), (1, 16, 16), (1, 40, 48), (1, 8, 12), (1, 32, 32), (1, 16, 24), (1, 24, 24), (1, 48, 48), (1, 8, 8), (1, 12, 16)]
    caps = [70, 130, 5, 200, 64, 90, 33, 120, 150, 1, 77]
    reqs = []
    for i, (g, cap) in enumerate(zip(grids, caps)):
        pv, grid, ids, lens = _batch(cfg, [g], 400 + i)
        reqs.append((ids, pv, grid, cap))
    mk = lambda dev: [Request(ids, torch.from_numpy(pv).cuda() if dev else pv, grid, cap) for ids, pv, grid, cap in reqs]
    ref = ContinuousBatcher(eng, chunk=8).run(mk(False))
    assert [len(r) for r in ref] == caps
    for k, dev in ((1, False), (1, True), (3, True), (4, False)):
        got = ContinuousBatcher(eng, chunk=8, prefetch=k).run(mk(dev))
        assert all(np.array_equal(a, b) for a, b in zip(ref, got)), (k, dev)
    assert eng.kv_pool_info()[0] == eng.kv_pool_info()[1]
