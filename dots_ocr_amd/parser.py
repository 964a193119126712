"""``DotsOCRParser`` on the MI355X engine.

Same constructor keywords, methods, on-disk outputs and result dicts as the reference class
(dots_ocr/parser.py:17-322); the difference is what sits behind ``_inference_with_hf``: instead of
HF ``AutoModelForCausalLM`` + flash-attn on CUDA (parser.py:62-117) the model object is
``DotsOcrHipForCausalLM`` (hand-written gfx950 kernels behind a C ABI) and the processor is
``DotsOcrProcessor``.  ``use_hf=True`` therefore selects the HIP engine.  PDF parsing feeds ALL pages
of a document to the engine at once (the reference forces one page at a time, parser.py:279-282): with the
real engine as a pipeline — host threads prepare pages ahead of the GPU, the continuous batcher keeps the
sequence slots full, finished pages are post-processed (layout clean-up, drawing, markdown, file writes:
parser.py:171-262) on host threads while the GPU decodes the others.
The vLLM HTTP client path (use_hf=False, model/inference.py) talks to an external server and is kept
for API compatibility only.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional

from .consts import MAX_PIXELS, MIN_PIXELS, image_extensions
from .format_transformer import layoutjson2md
from .image_utils import fetch_image, get_image_by_fitz_doc, smart_resize
from .layout_utils import draw_layout_on_image, post_process_output, pre_process_bboxes
from .prompts import dict_promptmode_to_prompt

_LAYOUT_MODES = ("prompt_layout_all_en", "prompt_layout_only_en", "prompt_grounding_ocr")


class DotsOCRParser:
    """parse image or pdf file"""

    def __init__(self, protocol="http", ip="localhost", port=8000, model_name="model", temperature=0.1, top_p=1.0,
                 max_completion_tokens=16384, num_thread=64, dpi=200, output_dir="./output", min_pixels=None,
                 max_pixels=None, use_hf=False, model_path="./weights/DotsOCR", model=None, processor=None,
                 hf_max_new_tokens=24000):
        self.dpi = dpi
        self.protocol, self.ip, self.port, self.model_name = protocol, ip, port, model_name
        self.temperature, self.top_p, self.max_completion_tokens = temperature, top_p, max_completion_tokens
        self.num_thread = num_thread
        self.output_dir = output_dir
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        self.model_path = model_path
        self.hf_max_new_tokens = hf_max_new_tokens
        self.use_hf = use_hf or model is not None
        if model is not None:                       # injected engine (tests, random-weight smoke runs)
            self.model, self.processor = model, processor
            from .processing import process_vision_info
            self.process_vision_info = process_vision_info
        elif self.use_hf:
            self._load_hf_model()
            print("use hf model, num_thread will be set to 1")
        else:
            print(f"use vllm model, num_thread will be set to {self.num_thread}")
        assert self.min_pixels is None or self.min_pixels >= MIN_PIXELS
        assert self.max_pixels is None or self.max_pixels <= MAX_PIXELS

    # ------------------------------------------------------------------ backend (the drop-in boundary)
    def _load_hf_model(self):
        from .modeling import DotsOcrHipForCausalLM
        from .processing import DotsOcrProcessor, process_vision_info
        self.model = DotsOcrHipForCausalLM.from_pretrained(self.model_path)
        self.processor = DotsOcrProcessor.from_pretrained(self.model_path, engine=self.model.engine)   # GPU image preprocessing
        self.process_vision_info = process_vision_info

    def _build_inputs(self, images, prompts):
        texts, flat = [], []
        for image, prompt in zip(images, prompts):
            messages = [{"role": "user", "content": [{"type": "image", "image": image}, {"type": "text", "text": prompt}]}]
            texts.append(self.processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True))
            imgs, _ = self.process_vision_info(messages)
            flat.extend(imgs)
        return self.processor(text=texts, images=flat, videos=None, padding=True, return_tensors="pt")

    def _inference_batch_with_hf(self, images, prompts) -> List[str]:
        inputs = self._build_inputs(images, prompts)
        generated = self.model.generate(**inputs, max_new_tokens=self.hf_max_new_tokens)
        trimmed = [out[len(inp):] for inp, out in zip(inputs.input_ids, generated)]
        return self.processor.batch_decode(trimmed, skip_special_tokens=True, clean_up_tokenization_spaces=False)

    def _inference_with_hf(self, image, prompt) -> str:
        return self._inference_batch_with_hf([image], [prompt])[0]

    def _inference_with_vllm(self, image, prompt):
        from dots_ocr.model.inference import inference_with_vllm
        return inference_with_vllm(image, prompt, model_name=self.model_name, protocol=self.protocol, ip=self.ip,
                                   port=self.port, temperature=self.temperature, top_p=self.top_p,
                                   max_completion_tokens=self.max_completion_tokens)

    # ------------------------------------------------------------------ per-page pipeline
    def get_prompt(self, prompt_mode, bbox=None, origin_image=None, image=None, min_pixels=None, max_pixels=None):
        prompt = dict_promptmode_to_prompt[prompt_mode]
        if prompt_mode == "prompt_grounding_ocr":
            assert bbox is not None
            box = pre_process_bboxes(origin_image, [bbox], input_width=image.width, input_height=image.height,
                                     min_pixels=min_pixels, max_pixels=max_pixels)[0]
            prompt = prompt + str(box)
        return prompt

    def _prepare(self, origin_image, prompt_mode, source, bbox, fitz_preprocess):
        min_pixels, max_pixels = self.min_pixels, self.max_pixels
        if prompt_mode == "prompt_grounding_ocr":
            min_pixels, max_pixels = min_pixels or MIN_PIXELS, max_pixels or MAX_PIXELS
        if min_pixels is not None:
            assert min_pixels >= MIN_PIXELS, f"min_pixels should >= {MIN_PIXELS}"
        if max_pixels is not None:
            assert max_pixels <= MAX_PIXELS, f"max_pixels should <= {MAX_PIXELS}"
        src = get_image_by_fitz_doc(origin_image, target_dpi=self.dpi) if (source == "image" and fitz_preprocess) else origin_image
        image = fetch_image(src, min_pixels=min_pixels, max_pixels=max_pixels)
        prompt = self.get_prompt(prompt_mode, bbox, origin_image, image, min_pixels=min_pixels, max_pixels=max_pixels)
        return image, prompt, min_pixels, max_pixels

    def _save(self, response, origin_image, image, prompt_mode, save_dir, save_name, page_idx, min_pixels, max_pixels):
        ih, iw = smart_resize(image.height, image.width)
        result = {"page_no": page_idx, "input_height": ih, "input_width": iw}
        j = os.path.join(save_dir, f"{save_name}.json")
        jpg = os.path.join(save_dir, f"{save_name}.jpg")
        md = os.path.join(save_dir, f"{save_name}.md")

        def write(path, text):
            with open(path, "w", encoding="utf-8") as f:
                f.write(text)

        if prompt_mode in _LAYOUT_MODES:
            cells, filtered = post_process_output(response, prompt_mode, origin_image, image, min_pixels=min_pixels, max_pixels=max_pixels)
            if filtered and prompt_mode != "prompt_layout_only_en":
                write(j, json.dumps(response, ensure_ascii=False))
                origin_image.save(jpg)
                write(md, cells)
                result.update({"layout_info_path": j, "layout_image_path": jpg, "md_content_path": md, "filtered": True})
                return result
            try:
                drawn = draw_layout_on_image(origin_image, cells)
            except Exception as e:
                print(f"Error drawing layout on image: {e}")
                drawn = origin_image
            write(j, json.dumps(cells, ensure_ascii=False))
            drawn.save(jpg)
            result.update({"layout_info_path": j, "layout_image_path": jpg})
            if prompt_mode != "prompt_layout_only_en":
                nohf = os.path.join(save_dir, f"{save_name}_nohf.md")
                write(md, layoutjson2md(origin_image, cells, text_key="text"))
                write(nohf, layoutjson2md(origin_image, cells, text_key="text", no_page_hf=True))
                result.update({"md_content_path": md, "md_content_nohf_path": nohf})
        else:
            origin_image.save(jpg)
            write(md, response)
            result.update({"layout_image_path": jpg, "md_content_path": md})
        return result

    def _parse_single_image(self, origin_image, prompt_mode, save_dir, save_name, source="image", page_idx=0, bbox=None,
                            fitz_preprocess=False):
        image, prompt, mn, mx = self._prepare(origin_image, prompt_mode, source, bbox, fitz_preprocess)
        response = self._inference_with_hf(image, prompt) if self.use_hf else self._inference_with_vllm(image, prompt)
        if source == "pdf":
            save_name = f"{save_name}_page_{page_idx}"
        return self._save(response, origin_image, image, prompt_mode, save_dir, save_name, page_idx, mn, mx)

    def parse_image(self, input_path, filename, prompt_mode, save_dir, bbox=None, fitz_preprocess=False):
        origin_image = fetch_image(input_path)
        result = self._parse_single_image(origin_image, prompt_mode, save_dir, filename, source="image", bbox=bbox,
                                          fitz_preprocess=fitz_preprocess)
        result["file_path"] = input_path
        return [result]

    def _parse_pages_pipelined(self, images, filename, prompt_mode, save_dir, input_path):
        """prepare (host threads) -> preprocess + admit (this thread, GPU) -> decode (GPU, continuous batching) -> post-process
        (host threads).  Only this thread talks to the engine (one handle per GPU, not thread-safe)."""
        from multiprocessing.pool import ThreadPool
        from .scheduler import ContinuousBatcher, Request
        from .modeling import resolve_sampling
        engine = self.model.engine
        # generate(**inputs, max_new_tokens=...) (parser.py:110): greedy unless the checkpoint's generation_config.json samples
        t, p_ = resolve_sampling(getattr(self.model, "generation_config", None))
        engine.set_sampling(t, p_, 0)
        eos = list(self.model.config.eos_token_ids)
        cb = ContinuousBatcher(engine, eos_ids=eos)
        n = len(images)
        results = [None] * n
        with ThreadPool(max(1, min(8, self.num_thread, n))) as prep_pool, ThreadPool(max(1, min(8, self.num_thread, n))) as post_pool:
            prep = [prep_pool.apply_async(self._prepare, (im, prompt_mode, "pdf", None, False)) for im in images]
            prepared, posts, nxt = {}, {}, 0
            while nxt < n or not cb.idle:
                # admit pages in order, without stalling the GPU on a page whose host preparation is still running
                while nxt < n and len(cb.pending) < cb.n_slots and (prep[nxt].ready() or cb.idle):
                    image, prompt, mn, mx = prepared[nxt] = prep[nxt].get()
                    inputs = self._build_inputs([image], [prompt])
                    ids = inputs["input_ids"][0]
                    ids = ids[inputs["attention_mask"][0].bool()].cpu().numpy()
                    cb.submit(Request(ids, inputs.get("pixel_values"), None if "image_grid_thw" not in inputs
                                      else inputs["image_grid_thw"].cpu().numpy(), self.hf_max_new_tokens, tag=nxt))
                    nxt += 1
                for _, req, toks in cb.step():
                    i = req.tag
                    toks = [int(t) for t in toks]
                    text = self.processor.batch_decode([toks], skip_special_tokens=True, clean_up_tokenization_spaces=False)[0]
                    image, _, mn, mx = prepared.pop(i)
                    posts[i] = post_pool.apply_async(self._save, (text, images[i], image, prompt_mode, save_dir, f"{filename}_page_{i}", i, mn, mx))
            for i, fut in posts.items():
                results[i] = fut.get()
                results[i]["file_path"] = input_path
        return results

    def parse_pages(self, images, filename, prompt_mode, save_dir, input_path=None):
        """All pages of one document through the engine at once — the caller-side half of page batching (SURVEY §8(f)
        rows 2 and 4): pipelined over the engine's sequence slots when the model has them, else ONE batched generate."""
        if self.use_hf and hasattr(getattr(self.model, "engine", None), "slots_prefill") and len(images) > 1:
            return self._parse_pages_pipelined(images, filename, prompt_mode, save_dir, input_path)
        prepared = [self._prepare(im, prompt_mode, "pdf", None, False) for im in images]
        if self.use_hf:
            responses = self._inference_batch_with_hf([p[0] for p in prepared], [p[1] for p in prepared])
        else:
            from multiprocessing.pool import ThreadPool
            with ThreadPool(max(1, min(len(images), self.num_thread))) as pool:
                responses = pool.starmap(self._inference_with_vllm, [(p[0], p[1]) for p in prepared])
        results = []
        for i, (im, p, resp) in enumerate(zip(images, prepared, responses)):
            r = self._save(resp, im, p[0], prompt_mode, save_dir, f"{filename}_page_{i}", i, p[2], p[3])
            r["file_path"] = input_path
            results.append(r)
        return results

    def parse_pdf(self, input_path, filename, prompt_mode, save_dir):
        from .doc_utils import load_images_from_pdf
        print(f"loading pdf: {input_path}")
        images = load_images_from_pdf(input_path, dpi=self.dpi)
        print(f"Parsing PDF with {len(images)} pages in one batch...")
        return self.parse_pages(images, filename, prompt_mode, save_dir, input_path=input_path)

    def parse_file(self, input_path, output_dir="", prompt_mode="prompt_layout_all_en", bbox=None, fitz_preprocess=False):
        output_dir = os.path.abspath(output_dir or self.output_dir)
        filename, ext = os.path.splitext(os.path.basename(input_path))
        save_dir = os.path.join(output_dir, filename)
        os.makedirs(save_dir, exist_ok=True)
        if ext == ".pdf":
            results = self.parse_pdf(input_path, filename, prompt_mode, save_dir)
        elif ext in image_extensions:
            results = self.parse_image(input_path, filename, prompt_mode, save_dir, bbox=bbox, fitz_preprocess=fitz_preprocess)
        else:
            raise ValueError(f"file extension {ext} not supported, supported extensions are {image_extensions} and pdf")
        print(f"Parsing finished, results saving to {save_dir}")
        with open(os.path.join(output_dir, os.path.basename(filename) + ".jsonl"), "w", encoding="utf-8") as w:
            for r in results:
                w.write(json.dumps(r, ensure_ascii=False) + "\n")
        return results


def main(argv=None):
    """The reference's command line (dots_ocr/parser.py:326-430), same arguments and defaults: `python dots_ocr/parser.py
    <pdf or image> [--use_hf true] ...`.  --use_hf selects the in-process MI355X engine, otherwise the OpenAI-compatible
    server at --protocol://--ip:--port (dots_ocr_amd.server, or any vLLM deployment)."""
    import argparse
    from .prompts import dict_promptmode_to_prompt
    ap = argparse.ArgumentParser(description="dots.ocr Multilingual Document Layout Parser")
    ap.add_argument("input_path", type=str, help="Input PDF/image file path")
    ap.add_argument("--output", type=str, default="./output", help="Output directory (default: ./output)")
    ap.add_argument("--prompt", choices=list(dict_promptmode_to_prompt.keys()), type=str, default="prompt_layout_all_en",
                    help="prompt to query the model, different prompts for different tasks")
    ap.add_argument("--bbox", type=int, nargs=4, metavar=("x1", "y1", "x2", "y2"), help="needed for prompt_grounding_ocr")
    ap.add_argument("--protocol", type=str, choices=["http", "https"], default="http")
    ap.add_argument("--ip", type=str, default="localhost")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--model_name", type=str, default="model")
    ap.add_argument("--temperature", type=float, default=0.1)
    ap.add_argument("--top_p", type=float, default=1.0)
    ap.add_argument("--dpi", type=int, default=200)
    ap.add_argument("--max_completion_tokens", type=int, default=16384)
    ap.add_argument("--num_thread", type=int, default=16)
    ap.add_argument("--no_fitz_preprocess", action="store_true",
                    help="skip the PDF round trip that re-renders low-dpi images at --dpi (needs PyMuPDF)")
    ap.add_argument("--min_pixels", type=int, default=None)
    ap.add_argument("--max_pixels", type=int, default=None)
    ap.add_argument("--use_hf", type=bool, default=False)
    a = ap.parse_args(argv)
    parser = DotsOCRParser(protocol=a.protocol, ip=a.ip, port=a.port, model_name=a.model_name, temperature=a.temperature, top_p=a.top_p,
                           max_completion_tokens=a.max_completion_tokens, num_thread=a.num_thread, dpi=a.dpi, output_dir=a.output,
                           min_pixels=a.min_pixels, max_pixels=a.max_pixels, use_hf=a.use_hf)
    fitz_preprocess = not a.no_fitz_preprocess
    if fitz_preprocess:
        print("Using fitz preprocess for image input, check the change of the image pixels")
    return parser.parse_file(a.input_path, prompt_mode=a.prompt, bbox=a.bbox, fitz_preprocess=fitz_preprocess)


if __name__ == "__main__":
    main()
