"""OpenAI-compatible client of an external vLLM server (reference dots_ocr/model/inference.py:7-48).
Out of the accelerated path — kept so `DotsOCRParser(use_hf=False)` still works."""
import os

from dots_ocr_amd.image_utils import PILimage_to_base64


def inference_with_vllm(image, prompt, protocol="http", ip="localhost", port=8000, temperature=0.1, top_p=0.9,
                        max_completion_tokens=32768, model_name="rednote-hilab/dots.mocr", system_prompt=None):
    import requests
    from openai import OpenAI        # optional dependency, imported lazily
    client = OpenAI(api_key=os.environ.get("API_KEY", "0"), base_url=f"{protocol}://{ip}:{port}/v1")
    messages = [{"role": "system", "content": system_prompt}] if system_prompt else []
    messages.append({"role": "user", "content": [
        {"type": "image_url", "image_url": {"url": PILimage_to_base64(image)}},
        {"type": "text", "text": f"<|img|><|imgpad|><|endofimg|>{prompt}"}]})
    try:
        r = client.chat.completions.create(messages=messages, model=model_name, max_completion_tokens=max_completion_tokens,
                                           temperature=temperature, top_p=top_p)
        return r.choices[0].message.content
    except requests.exceptions.RequestException as e:
        print(f"request error: {e}")
        return None
