"""OpenAI-compatible `/v1/chat/completions` endpoint on top of the HIP engine.

The reference's DEFAULT backend is an HTTP client of a vLLM server (dots_ocr/model/inference.py:7-48, used by
`DotsOCRParser(use_hf=False)`, demo/demo_vllm*.py and the Gradio/Streamlit apps): it POSTs one user message holding an
`image_url` data URL (base64 PNG, image_utils.py:67-71) and a text part "<|img|><|imgpad|><|endofimg|>{prompt}", with
`max_completion_tokens`, `temperature`, `top_p` (inference.py:28-43), and reads `choices[0].message.content`.  This module
serves exactly that wire format from the MI355X engine, so those callers work unchanged when pointed at it:

    python -m dots_ocr_amd.server --model-path ./weights/DotsOCR --port 8000

Concurrent requests (the reference client uses a ThreadPool of up to 64, parser.py:286-290) share the engine the way they
share a vLLM server: `ContinuousWorker` admits a request into a free sequence slot as soon as one exists and refills slots
as pages finish (dots_ocr_amd/scheduler.py), for requests with the same sampling parameters; requests with other
parameters wait for the running set to drain.  `BatchingWorker` (static batches through `model.generate`) remains for
model objects without engine slots.
"""
from __future__ import annotations

import argparse
import queue
import threading
import time
import uuid
from concurrent.futures import Future
from typing import List, Optional

from .image_utils import fetch_image
from .processing import ASSISTANT, END_USER, IMG_END, IMG_PAD, IMG_START, USER


class _Job:
    __slots__ = ("image", "text", "max_tokens", "temperature", "top_p", "future")

    def __init__(self, image, text, max_tokens, temperature, top_p):
        self.image, self.text, self.max_tokens, self.temperature, self.top_p = image, text, max_tokens, temperature, top_p
        self.future: Future = Future()


class BatchingWorker:
    """Single consumer thread (the engine handle is not thread-safe): groups queued jobs by sampling parameters."""

    def __init__(self, model, processor, max_batch: int = 8, max_wait_ms: float = 5.0, seed: int = 0):
        self.model, self.processor = model, processor
        self.max_batch, self.max_wait = max_batch, max_wait_ms / 1e3
        self.q: "queue.Queue[_Job]" = queue.Queue()
        self.seed = seed
        self.batches: List[int] = []            # sizes of the executed batches (observability / tests)
        self._stop = False
        self.thread = threading.Thread(target=self._run, name="dots-ocr-batcher", daemon=True)
        self.thread.start()

    def submit(self, job: _Job) -> Future:
        self.q.put(job)
        return job.future

    def close(self):
        self._stop = True
        self.q.put(None)
        self.thread.join(timeout=5)

    def _run(self):
        pending: List[_Job] = []
        while not self._stop:
            if not pending:
                job = self.q.get()
                if job is None:
                    return
                pending.append(job)
            deadline = time.monotonic() + self.max_wait
            while len(pending) < 4 * self.max_batch:
                try:
                    job = self.q.get(timeout=max(0.0, deadline - time.monotonic()))
                except queue.Empty:
                    break
                if job is None:
                    self._stop = True
                    break
                pending.append(job)
            key = (pending[0].max_tokens, pending[0].temperature, pending[0].top_p)
            batch = [j for j in pending if (j.max_tokens, j.temperature, j.top_p) == key][: self.max_batch]
            pending = [j for j in pending if j not in batch]
            self._execute(batch)

    def _execute(self, batch: List[_Job]):
        try:
            images = [j.image for j in batch if j.image is not None]
            inputs = self.processor(text=[j.text for j in batch], images=images or None, padding=True, return_tensors="pt")
            j0 = batch[0]
            self.seed += 1
            out = self.model.generate(**inputs, max_new_tokens=j0.max_tokens, do_sample=j0.temperature > 0,
                                      temperature=j0.temperature, top_p=j0.top_p, seed=self.seed)
            new = [o[len(i):] for i, o in zip(inputs.input_ids, out)]
            texts = self.processor.batch_decode(new, skip_special_tokens=True, clean_up_tokenization_spaces=False)
            pad = self.processor.tokenizer.pad_token_id
            eos = set(getattr(self.model, "config", None).eos_token_ids) if getattr(self.model, "config", None) else set()
            self.batches.append(len(batch))
            for j, t, ids, inp in zip(batch, texts, new, inputs.input_ids):
                toks = [int(x) for x in ids.tolist()]
                while toks and toks[-1] == pad and pad not in eos:
                    toks.pop()
                n_new = len(toks)
                hit_eos = any(x in eos for x in toks)
                j.future.set_result({"text": t, "prompt_tokens": int((inp != pad).sum()) if pad not in eos else int(len(inp)),
                                     "completion_tokens": n_new, "finish_reason": "stop" if hit_eos else "length"})
        except Exception as e:                      # surface the failure to every waiting request
            for j in batch:
                if not j.future.done():
                    j.future.set_exception(e)


class ContinuousWorker(BatchingWorker):
    """Continuous batching over the engine's sequence slots; same submit()/close() surface as BatchingWorker.
    `batches` records the number of occupied slots after every admission."""

    def __init__(self, model, processor, max_batch: int = 8, max_wait_ms: float = 5.0, seed: int = 0, chunk: int = 16,
                 look_ahead: Optional[int] = None):
        self.chunk = chunk
        # scheduler look-ahead (ContinuousBatcher(prefetch=k)): the towers of the next k queued requests run on the CU-masked side stream
        # beside the occupied slots.  Measured on A4 pages of mixed output length (tools/serve_bench.py, profiles/r04_serve_bench_a4_*.json):
        # 8 slots 3.26 -> 3.50 pages/s with k = 2, 16 slots 4.13 -> 4.36 with k = 8; identical tokens.  0 switches it off.
        self.look_ahead = (2 if max_batch <= 8 else min(8, max_batch // 2)) if look_ahead is None else max(0, int(look_ahead))
        super().__init__(model, processor, max_batch=max_batch, max_wait_ms=max_wait_ms, seed=seed)

    @staticmethod
    def _key(j: _Job):
        return (j.temperature, j.top_p)

    def _finish(self, job: _Job, prompt_tokens: int, toks, kv_truncated: bool = False):
        eos = set(self.model.config.eos_token_ids)
        toks = [int(t) for t in toks]
        text = self.processor.batch_decode([toks], skip_special_tokens=True, clean_up_tokenization_spaces=False)[0]
        # "kv_pool_exhausted": the engine ended the sequence at what its KV pages hold (vLLM would preempt and recompute; here the
        # caller sees that the output is short for a reason other than max_tokens and can resubmit)
        reason = "stop" if toks and toks[-1] in eos else ("kv_pool_exhausted" if kv_truncated else "length")
        job.future.set_result({"text": text, "prompt_tokens": prompt_tokens, "completion_tokens": len(toks), "finish_reason": reason})

    def _run(self):
        from collections import deque
        from .scheduler import ContinuousBatcher, Request
        engine = self.model.engine
        waiting: "deque[_Job]" = deque()
        cb, key = None, None
        while True:
            busy = cb is not None and not cb.idle
            if not busy and not waiting:
                job = self.q.get()                                   # nothing to do: block
                if job is None:
                    return
                waiting.append(job)
            while True:                                              # take whatever else has arrived
                try:
                    job = self.q.get_nowait()
                except queue.Empty:
                    break
                if job is None:
                    self._stop = True
                    break
                waiting.append(job)
            if self._stop and not busy and not waiting:
                return
            try:
                if not busy and waiting and (cb is None or key != self._key(waiting[0])):
                    key = self._key(waiting[0])                      # switch sampling parameters between drained sets only
                    self.seed += 1
                    engine.set_sampling(key[0], key[1], self.seed)
                    cb = ContinuousBatcher(engine, eos_ids=self.model.config.eos_token_ids, chunk=self.chunk, prefetch=self.look_ahead)
                # admit the FIFO prefix that shares the running parameters; a different request at the head makes the set drain
                admitted = 0
                while waiting and self._key(waiting[0]) == key and len(cb.pending) < 2 * cb.n_slots:
                    job = waiting.popleft()
                    try:
                        inputs = self.processor(text=[job.text], images=[job.image] if job.image is not None else None,
                                                padding=True, return_tensors="pt")
                        ids = inputs["input_ids"][0].numpy()
                        cb.submit(Request(ids, inputs.get("pixel_values"), None if "image_grid_thw" not in inputs
                                          else inputs["image_grid_thw"].numpy(), job.max_tokens, tag=job))
                        admitted += 1
                    except Exception as e:                           # a bad request fails alone
                        job.future.set_exception(e)
                if cb is not None and not cb.idle:
                    for _, req, toks in cb.step():
                        self._finish(req.tag, int(req.input_ids.shape[0]), toks, getattr(req, "kv_truncated", False))
                    if admitted:
                        self.batches.append(len(cb.running))
            except Exception as e:                                   # engine failure: fail everything in flight, start clean
                if cb is not None:
                    for _, req in list(cb.running.values()) + list(cb.pending) + list(cb._ahead):
                        if not req.tag.future.done():
                            req.tag.future.set_exception(e)
                cb, key = None, None


def _load_request_image(url: str, allow_remote: bool, allow_local: bool):
    """Image of a chat request.  Default: `data:image/...;base64,` URLs only — all the reference client sends
    (model/inference.py:20-33).  http(s) URLs and local paths are opt-in server flags: the server listens on 0.0.0.0, so
    resolving client-supplied locations would let a remote caller read local files or reach internal services (SSRF)."""
    if not isinstance(url, str):
        raise ValueError("image_url must be a string")
    if url.startswith("data:image"):
        return fetch_image(url)
    if url.startswith(("http://", "https://")):
        if not allow_remote:
            raise ValueError("remote image URLs are disabled (start the server with --allow-remote-images)")
        import requests
        from io import BytesIO
        from PIL import Image
        resp = requests.get(url, timeout=(3.0, 10.0), stream=True)
        resp.raise_for_status()
        data = resp.raw.read(64 * 1024 * 1024 + 1, decode_content=True)
        if len(data) > 64 * 1024 * 1024:
            raise ValueError("remote image larger than 64 MiB")
        return fetch_image(Image.open(BytesIO(data)))
    if not allow_local:
        raise ValueError("local image paths are disabled (start the server with --allow-local-images)")
    return fetch_image(url)


def _parse_messages(messages, processor=None, allow_remote: bool = False, allow_local: bool = False):
    """OpenAI chat messages -> (PIL image or None, chat-template text).  The text is rendered by the processor's chat
    template (the checkpoint's own Jinja template when it ships one).  The reference client writes the image placeholder
    tokens into its text part itself (model/inference.py:33); a client that sends them and an image part gets ONE image."""
    image, conv = None, []
    for m in messages:
        role, content = m.get("role", "user"), m.get("content")
        if isinstance(content, str):
            conv.append({"role": role, "content": [{"type": "text", "text": content}]})
            continue
        items = []
        for c in content or []:
            if c.get("type") == "image_url":
                url = c["image_url"]["url"] if isinstance(c["image_url"], dict) else c["image_url"]
                if image is not None:                  # one page per request, like the reference client (model/inference.py:23-43)
                    raise ValueError("exactly one image per request is supported")
                image = _load_request_image(url, allow_remote, allow_local)
                items.append({"type": "image", "image": "request"})
            elif c.get("type") == "text":
                items.append({"type": "text", "text": c["text"]})
        conv.append({"role": role, "content": items})
    for m in conv:                                    # placeholders already written into THIS message's text: drop its image item
        if any(IMG_PAD in it.get("text", "") for it in m["content"]):
            m["content"] = [it for it in m["content"] if it.get("type") != "image"]
    if processor is not None:
        return image, processor.apply_chat_template(conv, tokenize=False, add_generation_prompt=True)
    out = []                                          # no processor (unit tests): the stand-in template
    for m in conv:
        body = "".join(IMG_START + IMG_PAD + IMG_END if it.get("type") == "image" else it.get("text", "") for it in m["content"])
        out.append(body if m["role"] == "system" else (USER + body + END_USER if m["role"] == "user" else ASSISTANT + body))
    return image, "".join(out) + ASSISTANT


def create_app(model, processor, model_name: str = "model", max_batch: int = 8, max_wait_ms: float = 5.0, continuous: Optional[bool] = None,
               allow_remote_images: bool = False, allow_local_images: bool = False, look_ahead: Optional[int] = None):
    from fastapi import FastAPI, HTTPException
    from fastapi.concurrency import run_in_threadpool

    app = FastAPI(title="dots.ocr MI355X engine")
    if continuous is None:
        continuous = hasattr(model, "engine")
    worker = (ContinuousWorker(model, processor, max_batch=max_batch, max_wait_ms=max_wait_ms, look_ahead=look_ahead) if continuous
              else BatchingWorker(model, processor, max_batch=max_batch, max_wait_ms=max_wait_ms))
    app.state.worker = worker

    @app.get("/health")
    def health():
        return {"status": "ok"}

    @app.get("/v1/models")
    def models():
        return {"object": "list", "data": [{"id": model_name, "object": "model", "owned_by": "dots_ocr_amd"}]}

    @app.post("/v1/chat/completions")
    async def chat(req: dict):
        if req.get("stream"):
            raise HTTPException(400, "streaming is not supported")
        messages = req.get("messages")
        if not messages:
            raise HTTPException(400, "messages is required")
        try:            # decoding (and, when enabled, fetching) an image must not stall the event loop
            image, text = await run_in_threadpool(_parse_messages, messages, processor, allow_remote_images, allow_local_images)
        except Exception as e:
            raise HTTPException(400, f"bad message content: {e}")
        max_tokens = int(req.get("max_completion_tokens") or req.get("max_tokens") or 16384)
        temperature = float(req.get("temperature", 1.0) if req.get("temperature") is not None else 1.0)
        top_p = float(req.get("top_p", 1.0) if req.get("top_p") is not None else 1.0)
        fut = worker.submit(_Job(image, text, max_tokens, max(0.0, temperature), min(max(top_p, 1e-6), 1.0)))
        try:
            res = await run_in_threadpool(fut.result)
        except Exception as e:
            raise HTTPException(500, f"generation failed: {e}")
        return {
            "id": "chatcmpl-" + uuid.uuid4().hex, "object": "chat.completion", "created": int(time.time()),
            "model": req.get("model", model_name),
            "choices": [{"index": 0, "message": {"role": "assistant", "content": res["text"]}, "finish_reason": res["finish_reason"]}],
            "usage": {"prompt_tokens": res["prompt_tokens"], "completion_tokens": res["completion_tokens"],
                      "total_tokens": res["prompt_tokens"] + res["completion_tokens"]},
        }

    @app.on_event("shutdown")
    def _shutdown():
        worker.close()

    return app


def main(argv: Optional[List[str]] = None):
    ap = argparse.ArgumentParser(description="OpenAI-compatible server for the dots.ocr MI355X engine")
    ap.add_argument("--model-path", default="./weights/DotsOCR")
    ap.add_argument("--random-weights", action="store_true", help="seeded random weights (no checkpoint): plumbing tests only")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--served-model-name", default="model")
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--fp8-weights", action="store_true", help="quantise the linears to e4m3 (per-output-channel scale) at load time")
    ap.add_argument("--static-batching", action="store_true", help="static batches through model.generate instead of continuous batching")
    ap.add_argument("--look-ahead", type=int, default=None,
                    help="continuous batching: vision towers of the next N queued requests run on a CU partition beside the decoding slots "
                         "(default 2 up to 8 slots, min(8, slots / 2) above; 0 = off)")
    ap.add_argument("--allow-remote-images", action="store_true", help="let requests name http(s) image URLs (off: data: URLs only)")
    ap.add_argument("--allow-local-images", action="store_true", help="let requests name image paths on the server's file system")
    a = ap.parse_args(argv)
    import uvicorn
    from .modeling import DotsOcrHipForCausalLM
    from .processing import DotsOcrProcessor
    if a.random_weights:
        model = DotsOcrHipForCausalLM.from_random(device=a.device, max_batch=a.max_batch, fp8_weights=a.fp8_weights)
        proc = DotsOcrProcessor(model.config, engine=model.engine)
    else:
        model = DotsOcrHipForCausalLM.from_pretrained(a.model_path, device=a.device, max_batch=a.max_batch, fp8_weights=a.fp8_weights)
        proc = DotsOcrProcessor.from_pretrained(a.model_path, engine=model.engine)
    uvicorn.run(create_app(model, proc, a.served_model_name, a.max_batch, continuous=not a.static_batching,
                           allow_remote_images=a.allow_remote_images, allow_local_images=a.allow_local_images, look_ahead=a.look_ahead),
                host=a.host, port=a.port)


if __name__ == "__main__":
    main()
