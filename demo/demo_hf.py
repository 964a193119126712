#!/usr/bin/env python3
"""Twin of the reference's demo/demo_hf.py (BASELINE configs[0] flow) on the MI355X engine: every prompt mode over one image
through the same three HF-shaped calls the reference makes (processor.apply_chat_template / processor(...) / model.generate /
processor.batch_decode), with DotsOcrHipForCausalLM + DotsOcrProcessor in place of AutoModelForCausalLM + AutoProcessor.

    python demo/demo_hf.py [--model-path ./weights/DotsOCR] [--image demo/demo_image1.jpg] [--max-new-tokens 24000]
    python demo/demo_hf.py --random-weights --max-new-tokens 128        # no checkpoint offline: seeded random weights, a
                                                                        # synthetic 1700x2250 page (demo_image1.jpg's size)
"""
import argparse
import os
import sys
import time
from pathlib import Path

if "LOCAL_RANK" not in os.environ:
    os.environ["LOCAL_RANK"] = "0"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from dots_ocr.utils import dict_promptmode_to_prompt  # noqa: E402
from dots_ocr_amd.modeling import DotsOcrHipForCausalLM  # noqa: E402
from dots_ocr_amd.processing import DotsOcrProcessor, process_vision_info  # noqa: E402


def inference(image_path, prompt, model, processor, max_new_tokens=24000):
    messages = [{"role": "user", "content": [{"type": "image", "image": image_path}, {"type": "text", "text": prompt}]}]
    text = processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    image_inputs, video_inputs = process_vision_info(messages)
    inputs = processor(text=[text], images=image_inputs, videos=video_inputs, padding=True, return_tensors="pt")
    inputs = inputs.to("cuda")
    generated_ids = model.generate(**inputs, max_new_tokens=max_new_tokens)
    generated_ids_trimmed = [out_ids[len(in_ids):] for in_ids, out_ids in zip(inputs.input_ids, generated_ids)]
    output_text = processor.batch_decode(generated_ids_trimmed, skip_special_tokens=True, clean_up_tokenization_spaces=False)
    print(output_text)
    return output_text, int(inputs.input_ids.shape[1]), len(generated_ids_trimmed[0])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", default="./weights/DotsOCR")
    ap.add_argument("--image", default="demo/demo_image1.jpg")
    ap.add_argument("--max-new-tokens", type=int, default=24000)
    ap.add_argument("--random-weights", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="with --random-weights: the small-dims test model")
    a = ap.parse_args(argv)
    if a.random_weights:
        from dots_ocr_amd.config import DotsConfig
        from dots_ocr_amd.synthetic import synth_page
        cfg = DotsConfig.tiny(layers=2, v_layers=2) if a.tiny else DotsConfig()
        model = DotsOcrHipForCausalLM.from_random(cfg, max_batch=1, max_seq_len=8192 if a.tiny else 32768, max_patches=19824 * 2)
        processor = DotsOcrProcessor(cfg, engine=model.engine)
        image = a.image if Path(a.image).exists() else synth_page(1, (1700, 2250))
    else:
        model = DotsOcrHipForCausalLM.from_pretrained(a.model_path)
        processor = DotsOcrProcessor.from_pretrained(a.model_path, engine=model.engine)
        image = a.image
    rows = []
    for prompt_mode, prompt in dict_promptmode_to_prompt.items():
        print(f"prompt: {prompt}")
        t0 = time.perf_counter()
        _, n_in, n_out = inference(image, prompt, model, processor, a.max_new_tokens)
        rows.append((prompt_mode, n_in, n_out, time.perf_counter() - t0))
    for mode, n_in, n_out, dt in rows:
        print(f"{mode:24s} prompt {n_in:6d} tokens, {n_out:5d} new tokens, {dt:6.2f} s")
    return rows


if __name__ == "__main__":
    main()
