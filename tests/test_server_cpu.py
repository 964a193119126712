"""OpenAI-compatible endpoint with a stand-in model on CPU: wire format of the reference's
vLLM client (dots_ocr/model/inference.py:23-45), dynamic batching of concurrent requests, error handling."""
import json
import threading

import pytest
import torch

from dots_ocr_amd.config import DotsConfig
from dots_ocr_amd.image_utils import PILimage_to_base64
from dots_ocr_amd.processing import DotsOcrProcessor
from dots_ocr_amd.synthetic import synth_page

fastapi = pytest.importorskip("fastapi")
from fastapi.testclient import TestClient  # noqa: E402


class _EchoModel:
    """generate() appends '<n image tokens>|<sampling params>' as text tokens; records batch sizes."""

    def __init__(self, proc, cfg):
        self.proc, self.config, self.calls = proc, cfg, []
        self.lock = threading.Lock()

    def generate(self, input_ids=None, max_new_tokens=0, do_sample=False, temperature=1.0, top_p=1.0, **kw):
        with self.lock:
            self.calls.append((input_ids.shape[0], do_sample, temperature, top_p, max_new_tokens))
        rows = []
        for row in input_ids:
            n_img = int((row == self.config.image_token_id).sum())
            txt = f"{n_img}|{int(do_sample)}|{temperature:g}|{top_p:g}"
            rows.append(torch.tensor(self.proc.tokenizer.encode(txt) + [self.config.eos_token_ids[0]]))
        L = max(len(r) for r in rows)
        pad = self.config.pad_token_id
        new = torch.stack([torch.cat([r, torch.full((L - len(r),), pad)]) for r in rows])
        return torch.cat([input_ids, new], dim=1)


@pytest.fixture()
def client():
    from dots_ocr_amd.server import create_app
    cfg = DotsConfig.tiny()
    proc = DotsOcrProcessor(cfg)
    model = _EchoModel(proc, cfg)
    app = create_app(model, proc, model_name="model", max_batch=4, max_wait_ms=50)
    with TestClient(app) as c:
        yield c, model, cfg


def _payload(page, prompt="Extract the text content from this image.", **kw):
    # exactly what dots_ocr/model/inference.py:23-43 sends
    body = {"model": "model", "messages": [{"role": "user", "content": [
        {"type": "image_url", "image_url": {"url": PILimage_to_base64(page)}},
        {"type": "text", "text": f"<|img|><|imgpad|><|endofimg|>{prompt}"}]}],
        "max_completion_tokens": 64, "temperature": 0.1, "top_p": 0.9}
    body.update(kw)
    return body


def test_reference_client_wire_format(client):
    c, model, cfg = client
    page = synth_page(0, (140, 84))
    r = c.post("/v1/chat/completions", json=_payload(page))
    assert r.status_code == 200, r.text
    d = r.json()
    assert d["object"] == "chat.completion" and d["choices"][0]["message"]["role"] == "assistant"
    n_img, sampled, temp, top_p = d["choices"][0]["message"]["content"].split("|")
    assert int(n_img) == (84 // 14) * (140 // 14) // 4 and sampled == "1" and float(temp) == 0.1 and float(top_p) == 0.9
    assert d["choices"][0]["finish_reason"] == "stop" and d["usage"]["completion_tokens"] > 0
    assert c.get("/v1/models").json()["data"][0]["id"] == "model" and c.get("/health").json() == {"status": "ok"}
    # temperature 0 -> greedy
    d0 = c.post("/v1/chat/completions", json=_payload(page, temperature=0)).json()
    assert d0["choices"][0]["message"]["content"].split("|")[1] == "0"
    # text-only request and a placeholder-less image request both work
    t = c.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": "hi"}], "max_tokens": 8}).json()
    assert t["choices"][0]["message"]["content"].startswith("0|")
    body = _payload(page)
    body["messages"][0]["content"][1]["text"] = "no placeholder"
    assert int(c.post("/v1/chat/completions", json=body).json()["choices"][0]["message"]["content"].split("|")[0]) == 15


def test_concurrent_requests_are_batched(client):
    c, model, cfg = client
    page = synth_page(1, (140, 84))
    results = [None] * 6

    def go(i):
        results[i] = c.post("/v1/chat/completions", json=_payload(page)).status_code
    th = [threading.Thread(target=go, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert results == [200] * 6
    sizes = [n for (n, *_rest) in model.calls]
    assert sum(sizes) == 6 and max(sizes) <= 4 and len(sizes) < 6          # grouped, never above max_batch


def test_bad_requests(client):
    c, _, _ = client
    assert c.post("/v1/chat/completions", json={"messages": []}).status_code == 400
    assert c.post("/v1/chat/completions", json=_payload(synth_page(0, (56, 56)), stream=True)).status_code == 400
    bad = {"messages": [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": "data:image/png;base64,@@@"}}]}]}
    assert c.post("/v1/chat/completions", json=bad).status_code == 400


class _SlotModel:
    """Model object with engine slots (stand-in engine from tests/fakes.py): the server picks ContinuousWorker for it."""

    def __init__(self, proc, cfg):
        from fakes import FakeSlotEngine
        self.config = cfg
        self.samplings = []

        def script(prompt):
            n_img = int((prompt == cfg.image_token_id).sum())
            t, p, _ = self.engine.sampling
            return proc.tokenizer.encode(f"{n_img}|{t:g}|{p:g}|" + "x" * (n_img % 7)) + [cfg.eos_token_ids[0]]
        self.engine = FakeSlotEngine(script, max_batch=3, max_patches=4096, max_prefill_tokens=4096, max_seq_len=2048)


def test_continuous_worker_keeps_slots_full_and_separates_sampling_parameters():
    from dots_ocr_amd.server import ContinuousWorker, create_app
    cfg = DotsConfig.tiny()
    proc = DotsOcrProcessor(cfg)
    model = _SlotModel(proc, cfg)
    app = create_app(model, proc, model_name="model", max_batch=3)
    with TestClient(app) as c:
        assert isinstance(app.state.worker, ContinuousWorker)
        app.state.worker.chunk = 2
        sizes = [(140, 84), (56, 56), (112, 84), (84, 140), (56, 112), (168, 56), (112, 112), (56, 84)]
        results = [None] * len(sizes)

        def go(i):
            kw = {"temperature": 0} if i % 4 == 3 else {}
            results[i] = c.post("/v1/chat/completions", json=_payload(synth_page(i, sizes[i]), **kw))
        th = [threading.Thread(target=go, args=(i,)) for i in range(len(sizes))]
        [t.start() for t in th]
        [t.join() for t in th]
        for i, r in enumerate(results):
            assert r.status_code == 200, r.text
            d = r.json()
            n_img, temp, top_p, _ = d["choices"][0]["message"]["content"].split("|")
            w, h = sizes[i]
            assert int(n_img) == (h // 14) * (w // 14) // 4
            assert (float(temp), float(top_p)) == ((0.0, 0.9) if i % 4 == 3 else (0.1, 0.9))     # each ran under its own parameters
            assert d["choices"][0]["finish_reason"] == "stop" and d["usage"]["prompt_tokens"] > int(n_img)
        eng = model.engine
        pre = [e[1] for e in eng.log if e[0] == "prefill"]
        assert sum(len(p) for p in pre) == len(sizes) and all(len(p) <= 3 for p in pre)
        assert max(app.state.worker.batches) <= 3 and not eng.slots
        # a capped request reports "length"
        d = c.post("/v1/chat/completions", json=_payload(synth_page(0, sizes[0]), max_completion_tokens=3)).json()
        assert d["choices"][0]["finish_reason"] == "length" and d["usage"]["completion_tokens"] == 3


def test_continuous_worker_look_ahead_default_and_override():
    """Round 4: the scheduler look-ahead is on by default in the server (2 requests up to 8 slots, min(8, slots / 2) above; measured in
    tools/serve_bench.py), `--look-ahead 0` / look_ahead=0 switches it off, and an engine without vit_prefetch silently runs without it."""
    from dots_ocr_amd.scheduler import ContinuousBatcher
    from dots_ocr_amd.server import ContinuousWorker
    cfg = DotsConfig.tiny()
    proc = DotsOcrProcessor(cfg)
    model = _SlotModel(proc, cfg)
    for slots, want in ((3, 2), (8, 2), (16, 8), (32, 8), (12, 6)):
        w = ContinuousWorker(model, proc, max_batch=slots)
        try:
            assert w.look_ahead == want
        finally:
            w.close()
    w = ContinuousWorker(model, proc, max_batch=8, look_ahead=0)
    try:
        assert w.look_ahead == 0
    finally:
        w.close()
    cb = ContinuousBatcher(model.engine, eos_ids=(), prefetch=4)
    assert cb.prefetch == (4 if hasattr(model.engine, "vit_prefetch") else 0)


def test_server_refuses_remote_and_local_image_locations_by_default(tmp_path):
    """a client-supplied image_url must not make the server read local files or fetch URLs (SSRF) unless the
    operator opted in; data: URLs (all the reference client sends, model/inference.py:20-33) always work."""
    from PIL import Image
    from dots_ocr_amd.image_utils import PILimage_to_base64
    from dots_ocr_amd.server import _parse_messages
    p = tmp_path / "secret.png"
    Image.new("RGB", (32, 32), "red").save(p)

    def msg(url):
        return [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": url}}, {"type": "text", "text": "read"}]}]
    for url in (str(p), "file://" + str(p), "http://169.254.169.254/latest/meta-data", "https://example.invalid/x.png"):
        with pytest.raises(ValueError):
            _parse_messages(msg(url))
    img, text = _parse_messages(msg(PILimage_to_base64(Image.new("RGB", (28, 28), "blue"))))
    assert img.size == (28, 28) and text.count("<|imgpad|>") == 1 and text.endswith("<|assistant|>")
    img, _ = _parse_messages(msg(str(p)), allow_local=True)
    assert img.size == (32, 32)
    # the reference client puts the placeholder tokens into its text itself: still exactly one image slot
    own = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": PILimage_to_base64(Image.new("RGB", (28, 28)))}},
                                         {"type": "text", "text": "<|img|><|imgpad|><|endofimg|>read"}]}]
    assert _parse_messages(own)[1].count("<|imgpad|>") == 1


def test_server_rejects_a_second_image_and_keeps_other_messages_images():
    """two image_url parts used to render two placeholder blocks for ONE kept image (an opaque token-count mismatch later);
    and a placeholder written into one message's text must not strip the image item of another message."""
    from PIL import Image
    from dots_ocr_amd.image_utils import PILimage_to_base64
    from dots_ocr_amd.server import _parse_messages
    u = PILimage_to_base64(Image.new("RGB", (28, 28), "blue"))
    two = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": u}}, {"type": "image_url", "image_url": {"url": u}},
                                        {"type": "text", "text": "read"}]}]
    with pytest.raises(ValueError, match="exactly one image"):
        _parse_messages(two)
    mixed = [{"role": "system", "content": [{"type": "text", "text": "about <|imgpad|> tokens"}]},
             {"role": "user", "content": [{"type": "image_url", "image_url": {"url": u}}, {"type": "text", "text": "read"}]}]
    img, text = _parse_messages(mixed)
    assert img is not None and text.count("<|img|><|imgpad|><|endofimg|>") == 1
