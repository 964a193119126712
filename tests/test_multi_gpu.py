"""N > 1 on hardware (VERDICT r2 missing #2 / next #3): the data-parallel page job, launched the way the driver launches bench.py
(`python -m torch.distributed.run --nproc-per-node 2`), one engine per rank, RCCL for the result gather — gathered tokens must equal
the single-rank run of the same pages (pages are independent units: reference dots_ocr/parser.py:265-277, results re-sorted by page
number at :292).  The RCCL test skips itself below 2 devices; the same worker also runs with both ranks on ONE GPU over gloo, so every
line but the RCCL transport is exercised on a 1-GPU box.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(n, out, extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "mp_dp_worker.py"), str(out)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(Path(out).read_text())


@pytest.fixture(scope="module")
def single_rank_tokens():
    sys.path.insert(0, str(ROOT / "tests"))
    import mp_dp_worker as w
    cfg, eng = w.make_engine(0)
    pages = w.job(cfg)
    ids, lens = w.run_shard(eng, pages, list(range(len(pages))))
    eng.close()
    want = [[p, ids[p, :lens[p]].tolist()] for p in range(len(pages))]
    assert [len(t) for _, t in want] == w.CAPS
    return want


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL over xGMI)")
def test_two_ranks_rccl_gather_equals_single_rank(single_rank_tokens, tmp_path):
    res = _launch(2, tmp_path / "rccl.json", {})
    assert res["world"] == 2 and res["rccl_ranks"] == 2 and res["backend"] == "nccl"
    assert min(res["pages_per_rank"]) >= 1 and sum(res["pages_per_rank"]) == len(single_rank_tokens)
    assert res["gathered"] == single_rank_tokens


@pytest.mark.skipif(torch.cuda.device_count() < 4, reason="needs 4 GPUs")
def test_four_ranks_rccl_gather_equals_single_rank(single_rank_tokens, tmp_path):
    res = _launch(4, tmp_path / "rccl4.json", {})
    assert res["world"] == 4 and res["rccl_ranks"] == 4
    assert res["gathered"] == single_rank_tokens


def test_two_ranks_on_one_gpu_gloo_gather_equals_single_rank(single_rank_tokens, tmp_path):
    """The same worker under torch.distributed.run with both ranks on cuda:0 and the gloo backend: sharding, per-rank continuous batching on
    the real engine, the two all_gathers and the page-order merge — everything of the N > 1 path except the RCCL transport."""
    res = _launch(2, tmp_path / "gloo.json", {"DOTS_TEST_DP_BACKEND": "gloo", "DOTS_TEST_DP_ONE_GPU": "1"})
    assert res["world"] == 2 and res["rccl_ranks"] == 2 and res["backend"] == "gloo"
    assert res["gathered"] == single_rank_tokens
