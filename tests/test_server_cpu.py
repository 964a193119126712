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
