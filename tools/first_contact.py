#!/usr/bin/env python3
"""First contact with a REAL dots.ocr checkpoint (the only route from "parity partial" to "green").

    python tools/first_contact.py --model-path ./weights/DotsOCR [--image demo/demo_image1.jpg] [--prompt-mode prompt_layout_all_en]
                                  [--steps 128] [--oracle-max-pixels 1000000] [--out profiles/first_contact.json] [--no-gpu]

Everything this repository believes about the checkpoint is [RECALLED]: the hub's modeling_dots_vision.py /
modeling_dots_ocr.py are not in /root/reference, tools/download_model.py needs a network, and no weights have ever been loaded
(reference call site: dots_ocr/parser.py:67-75).  This script turns the first time weights and a GPU are both present into ONE command
whose report says exactly where recollection and reality differ:

 1. config diff      config.json (+ vision_config), preprocessor_config.json, generation_config.json against SURVEY §8(a)'s [RECALLED]
                     table and against what DotsConfig.from_pretrained actually reads; keys the engine IGNORES that would change the
                     arithmetic (activation, rope scaling, sliding window, causal vision attention, biases ...) are flagged as blocking.
 2. tensor inventory names / shapes / dtypes in the *.safetensors headers (no weights are read) against weights.expected_tensors(cfg):
                     required-but-missing, present-but-unused (what the engine would silently not consume), shape mismatches.
 3. text side        tokenizer.json / chat template present?  The rendered prompt and its token count for the chosen prompt mode.
 4. the run          --image + prompt through the processor, then the bf16-emulated CPU oracle and the HIP engine, N greedy tokens: the engine
                     decodes step by step (fp32 logits of every step through the C ABI), the oracle is teacher-forced on the engine's tokens;
                     per step: token equality, the oracle's top-2 margin, max |logit error|; plus the decoded text of both.
                     The CPU oracle needs ~150 TFLOP for an A4 page, so by default the image is bounded to --oracle-max-pixels for this
                     comparison (0 = keep the reference's own smart_resize result).  --no-gpu runs stages 1-3 and the oracle alone
                     (CPU containers; tests/test_host_cpu.py drives it against a synthetic checkpoint directory).
Exit status: 0 = every stage ran and nothing blocking was found; 2 = blocking differences (the report lists them); 1 = a stage crashed.
"""
from __future__ import annotations

import argparse
import json
import struct
import sys
import time
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

# SURVEY.md §8(a), "[RECALLED] config.json the table above assumes (verify on first contact)"
RECALLED = {
    "config.json": {"hidden_size": 1536, "num_hidden_layers": 28, "num_attention_heads": 12, "num_key_value_heads": 2, "intermediate_size": 8960,
                    "vocab_size": 151936, "rope_theta": 1000000.0, "rms_norm_eps": 1e-6, "attention_bias": True, "tie_word_embeddings": False,
                    "max_position_embeddings": 131072, "image_token_id": 151665},
    "config.json:vision_config": {"embed_dim": 1536, "num_hidden_layers": 42, "num_attention_heads": 12, "intermediate_size": 4224, "patch_size": 14,
                                  "spatial_merge_size": 2, "temporal_patch_size": 1, "num_channels": 3, "rms_norm_eps": 1e-5, "use_bias": False,
                                  "post_norm": True, "is_causal": False},
    "preprocessor_config.json": {"min_pixels": 3136, "max_pixels": 11289600, "image_mean": [0.48145466, 0.4578275, 0.40821073],
                                 "image_std": [0.26862954, 0.26130258, 0.27577711]},
    "generation_config.json": {"eos_token_id": [151643, 151673]},
}
# keys DotsConfig.from_pretrained reads (dots_ocr_amd/config.py); anything else in the files is ignored by the engine
READ = {
    "config.json": {"hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim", "intermediate_size", "vocab_size", "rope_theta",
                    "rms_norm_eps", "attention_bias", "tie_word_embeddings", "max_position_embeddings", "image_token_id", "eos_token_id", "pad_token_id",
                    "vision_config"},
    "config.json:vision_config": {"embed_dim", "num_hidden_layers", "num_attention_heads", "intermediate_size", "patch_size", "spatial_merge_size",
                                  "temporal_patch_size", "num_channels", "rms_norm_eps", "use_bias", "post_norm", "hidden_size"},
    "preprocessor_config.json": {"min_pixels", "max_pixels", "image_mean", "image_std"},
    "generation_config.json": {"eos_token_id", "do_sample", "temperature", "top_p"},
}
# ignored keys whose value, if not the harmless one, means the engine computes something else than the checkpoint's own code
ARITHMETIC = {
    "config.json": {"hidden_act": ("silu",), "rope_scaling": (None,), "use_sliding_window": (False, None), "sliding_window": (None,), "attention_dropout": (0.0, 0, None),
                    "mlp_bias": (False, None), "use_mrope": (False, None), "layer_types": (None,)},
    "config.json:vision_config": {"is_causal": (False, None), "hidden_act": ("silu", None), "init_merger_std": None, "attn_implementation": None, "gradient_checkpointing": None},
    "preprocessor_config.json": {"do_resize": (True, None), "do_rescale": (True, None), "do_normalize": (True, None), "rescale_factor": (1 / 255, 0.00392156862745098, None),
                                 "patch_size": (14, None), "merge_size": (2, None), "temporal_patch_size": (1, None), "resample": (3, None), "do_convert_rgb": (True, None)},
}


def _same(a, b) -> bool:
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, bool) or isinstance(b, bool):
        return a == b
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return abs(a - b) <= 1e-9 * max(1.0, abs(a), abs(b))
    return a == b


def config_diff(path: Path) -> dict:
    files = {}
    for name in ("config.json", "preprocessor_config.json", "generation_config.json"):
        f = path / name
        files[name] = json.loads(f.read_text()) if f.exists() else None
    files["config.json:vision_config"] = (files["config.json"] or {}).get("vision_config")
    rows, ignored, blocking = [], [], []
    for sect, recalled in RECALLED.items():
        have = files.get(sect)
        if have is None:
            (blocking if sect.startswith("config.json") else ignored).append(f"{sect}: not present")
            continue
        for k, rv in recalled.items():
            if k not in have:
                rows.append({"where": sect, "key": k, "recalled": rv, "checkpoint": None, "status": "absent in the checkpoint (the recalled default applies)"})
            else:
                cv = have[k]
                if sect == "generation_config.json" and isinstance(cv, int):
                    cv = [cv]
                ok = _same(rv, cv)
                rows.append({"where": sect, "key": k, "recalled": rv, "checkpoint": have[k], "status": "same" if ok else "DIFFERENT"})
                if not ok and k in ("is_causal",):
                    blocking.append(f"{sect}: {k} = {have[k]!r}: the engine's vision attention is bidirectional")
        for k, v in have.items():
            if k in READ.get(sect, set()) or k in recalled:
                continue
            harmless = ARITHMETIC.get(sect, {}).get(k, "unknown")
            if harmless == "unknown":
                ignored.append({"where": sect, "key": k, "value": v if not isinstance(v, (dict, list)) or len(str(v)) < 200 else "<%d chars>" % len(str(v)), "effect": "not read by the engine"})
            elif harmless is not None and not any(_same(v, h) for h in harmless):
                blocking.append(f"{sect}: {k} = {v!r} is ignored by the engine, which computes as if it were {harmless[0]!r}")
    differing = [r for r in rows if r["status"] == "DIFFERENT"]
    return {"recalled_vs_checkpoint": rows, "differing_recalled_values": differing,
            "note_on_differences": "a DIFFERENT dimension is read from the checkpoint (DotsConfig.from_pretrained), so it is not an error by itself: it falsifies SURVEY §8(a) "
                                   "and every size-dependent claim (flops, bytes, tile shapes tuned for 1536 / 4224 / 8960) built on it",
            "keys_the_engine_ignores": ignored, "blocking": blocking}


def safetensors_headers(path: Path) -> dict:
    out = {}
    for f in sorted(path.glob("*.safetensors")):
        with open(f, "rb") as fh:
            n = struct.unpack("<Q", fh.read(8))[0]
            head = json.loads(fh.read(n))
        for name, meta in head.items():
            if name != "__metadata__":
                out[name] = {"shape": tuple(meta["shape"]), "dtype": meta["dtype"], "file": f.name}
    return out


def tensor_inventory(path: Path, cfg) -> dict:
    from dots_ocr_amd.weights import expected_tensors
    have, want = safetensors_headers(path), expected_tensors(cfg)
    missing = sorted(k for k in want if k not in have)
    unused = sorted(k for k in have if k not in want)
    mism = [{"name": k, "engine_expects": list(want[k]), "checkpoint": list(have[k]["shape"])} for k in want if k in have and tuple(have[k]["shape"]) != tuple(want[k])]
    dtypes = {}
    for m in have.values():
        dtypes[m["dtype"]] = dtypes.get(m["dtype"], 0) + 1
    prefixes = {}
    for k in unused:
        p = ".".join(k.split(".")[:3])
        prefixes[p] = prefixes.get(p, 0) + 1
    blocking = []
    if missing:
        blocking.append(f"{len(missing)} tensors the engine requires are absent (first: {missing[:3]})")
    if mism:
        blocking.append(f"{len(mism)} tensors have another shape than the engine derives from config.json (first: {mism[0]})")
    if unused:
        blocking.append(f"{len(unused)} checkpoint tensors would NOT be consumed by the engine (prefixes: {dict(sorted(prefixes.items(), key=lambda kv: -kv[1])[:8])}): "
                        "parameters the reference's forward uses and this engine does not model, or a naming difference")
    return {"tensors_in_checkpoint": len(have), "tensors_the_engine_consumes": len(want), "dtypes": dtypes, "required_but_missing": missing[:200],
            "present_but_unused": unused[:200], "unused_by_prefix": prefixes, "shape_mismatches": mism[:50], "blocking": blocking}


def text_side(path: Path, cfg, prompt_mode: str) -> dict:
    from dots_ocr_amd.processing import DotsOcrProcessor
    from dots_ocr_amd.prompts import dict_promptmode_to_prompt
    present = {n: (path / n).exists() for n in ("tokenizer.json", "tokenizer_config.json", "chat_template.json", "chat_template.jinja", "vocab.json", "merges.txt", "special_tokens_map.json")}
    proc = DotsOcrProcessor.from_pretrained(path)
    messages = [{"role": "user", "content": [{"type": "image", "image": "page"}, {"type": "text", "text": dict_promptmode_to_prompt[prompt_mode]}]}]
    text = proc.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    blocking = []
    if not present["tokenizer.json"]:
        blocking.append("tokenizer.json absent: the processor falls back to the synthetic byte tokenizer, token ids will not be the checkpoint's")
    return {"files": present, "tokenizer_class": type(proc.tokenizer).__name__, "rendered_prompt_head": text[:120], "rendered_prompt_tail": text[-80:],
            "prompt_tokens_without_image_pads": len(proc.tokenizer.encode(text)), "blocking": blocking}, proc


def bound_image(img, max_pixels: int):
    if max_pixels and img.width * img.height > max_pixels:
        s = (max_pixels / (img.width * img.height)) ** 0.5
        img = img.resize((max(28, int(img.width * s)), max(28, int(img.height * s))))
    return img


def run(path: Path, cfg, proc, image_path: Path, prompt_mode: str, steps: int, oracle_max_pixels: int, gpu: bool) -> dict:
    import numpy as np
    import torch
    from PIL import Image
    from dots_ocr_amd.prompts import dict_promptmode_to_prompt
    from dots_ocr_amd.weights import load_state_dict
    from oracle import model as om
    img = Image.open(image_path).convert("RGB")
    full = (img.width, img.height)
    img = bound_image(img, oracle_max_pixels)
    prompt = dict_promptmode_to_prompt[prompt_mode]
    messages = [{"role": "user", "content": [{"type": "image", "image": img}, {"type": "text", "text": prompt}]}]
    text = proc.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    inputs = proc(text=[text], images=[img], padding=True, return_tensors="pt")
    ids, pv, grid = inputs["input_ids"][0], inputs["pixel_values"], inputs["image_grid_thw"]
    rep = {"image": str(image_path), "image_size": full, "size_used": (img.width, img.height), "patches": int(pv.shape[0]), "prompt_tokens": int(ids.shape[0]),
           "prompt_mode": prompt_mode, "steps": steps,
           "note": None if (img.width, img.height) == full else f"image bounded to {oracle_max_pixels} pixels for the CPU oracle (--oracle-max-pixels 0 keeps the reference's size)"}
    sd = load_state_dict(path)
    eos = tuple(cfg.eos_token_ids)
    eng_tokens = eng_logits = None
    if gpu:
        from dots_ocr_amd.engine import Engine
        eng = Engine(cfg, max_batch=1, max_seq_len=int(ids.shape[0]) + steps + 64, max_patches=int(pv.shape[0]) + 64, max_prefill_tokens=int(ids.shape[0]) + 64)
        eng.load_state_dict(sd)
        t0 = time.perf_counter()
        eng.vit_forward(pv.numpy(), grid.numpy())
        eng.prefill(ids.numpy().astype(np.int32), np.asarray([ids.shape[0]], np.int32))
        eng_logits, eng_tokens = [eng.get_logits()[0].copy()], [int(eng.get_last_tokens()[0])]
        while len(eng_tokens) < steps and eng_tokens[-1] not in eos:
            eng.decode_step()
            eng_logits.append(eng.get_logits()[0].copy())
            eng_tokens.append(int(eng.get_last_tokens()[0]))
        eng.synchronize()
        rep["engine_seconds"] = time.perf_counter() - t0
        rep["engine_text"] = proc.batch_decode([eng_tokens])[0]
        eng.close()
    sd32 = {k: v.float() for k, v in sd.items()}
    t0 = time.perf_counter()
    n = len(eng_tokens) if eng_tokens else steps
    otoks, olog = om.generate(sd32, cfg, ids, pv, grid, n, eos_ids=() if eng_tokens else eos, emulate_bf16=True, forced_tokens=eng_tokens, return_logits=True)
    rep["oracle_seconds"] = time.perf_counter() - t0
    rep["oracle_text"] = proc.batch_decode([otoks])[0]
    rep["oracle_tokens"] = [int(t) for t in otoks]
    if eng_tokens:
        rows, prefix, broke = [], 0, False
        for s, (tok, el, ol) in enumerate(zip(eng_tokens, eng_logits, olog)):
            top2 = torch.topk(ol, 2)
            err = float((torch.from_numpy(el).double() - ol.double()).abs().max())
            same = tok == int(top2.indices[0])
            if same and not broke:
                prefix += 1
            broke = broke or not same
            rows.append({"step": s, "engine_token": tok, "oracle_argmax": int(top2.indices[0]), "token_equal": same,
                         "oracle_top2_margin": float(top2.values[0] - top2.values[1]), "max_abs_logit_err": err, "logit_range": float(ol.max() - ol.min())})
        rep.update({"engine_tokens": eng_tokens, "tokens_equal": sum(int(r["token_equal"]) for r in rows), "token_exact_prefix": prefix,
                    "violations_outside_near_tie_band": [r for r in rows if not r["token_equal"] and r["oracle_top2_margin"] > 2 * r["max_abs_logit_err"]][:8],
                    "max_abs_logit_err": max(r["max_abs_logit_err"] for r in rows), "per_step": rows,
                    "rule": "token == bf16-emulated oracle arg max unless the oracle's top-2 margin <= 2 x that step's max |logit error| (the suite's rule)"})
    return rep


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--image", default=str(ROOT / "demo" / "demo_image1.jpg"))
    ap.add_argument("--prompt-mode", default="prompt_layout_all_en")
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--oracle-max-pixels", type=int, default=1_000_000)
    ap.add_argument("--out", default=str(ROOT / "profiles" / "first_contact.json"))
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    path = Path(a.model_path)
    rep = {"model_path": str(path), "stages": {}}
    blocking, crashed = [], []

    def stage(name, fn):
        try:
            r = fn()
            rep["stages"][name] = r[0] if isinstance(r, tuple) else r
            blocking.extend(f"[{name}] {b}" for b in rep["stages"][name].get("blocking", []))
            return r
        except Exception as e:                     # keep going: a later stage may still be informative
            rep["stages"][name] = {"crashed": repr(e), "traceback": traceback.format_exc()[-1500:]}
            crashed.append(name)
            return None
    stage("1_config_diff", lambda: config_diff(path))
    from dots_ocr_amd.config import DotsConfig
    cfg = None
    try:
        cfg = DotsConfig.from_pretrained(path)
    except Exception as e:
        rep["stages"]["config_load"] = {"crashed": repr(e)}
        crashed.append("config_load")
    proc = None
    if cfg is not None:
        stage("2_tensor_inventory", lambda: tensor_inventory(path, cfg))
        r = stage("3_text_side", lambda: text_side(path, cfg, a.prompt_mode))
        proc = r[1] if r else None
    if cfg is not None and proc is not None:
        gpu = not a.no_gpu
        if gpu:
            import torch
            if not torch.cuda.is_available():
                raise SystemExit("first_contact: no GPU visible; the engine has no CPU fallback (pass --no-gpu for stages 1-3 + the oracle alone)")
        stage("4_run", lambda: run(path, cfg, proc, Path(a.image), a.prompt_mode, a.steps, a.oracle_max_pixels, gpu))
        r4 = rep["stages"].get("4_run", {})
        if r4.get("violations_outside_near_tie_band"):
            blocking.append(f"[4_run] {len(r4['violations_outside_near_tie_band'])} engine tokens differ from the oracle outside the near-tie band")
    rep["blocking"], rep["crashed_stages"] = blocking, crashed
    rep["verdict"] = "crashed" if crashed else ("blocking differences" if blocking else "clean: the recalled architecture matches, every tensor is consumed, tokens follow the oracle")
    out = Path(a.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps(rep, indent=1, default=str))
    brief = {k: v for k, v in rep.items() if k != "stages"}
    brief["stage_summaries"] = {k: {kk: vv for kk, vv in v.items() if kk in ("crashed", "tokens_equal", "token_exact_prefix", "max_abs_logit_err", "tensors_in_checkpoint", "tensors_the_engine_consumes",
                                                                             "engine_text", "oracle_text", "tokenizer_class")} for k, v in rep["stages"].items()}
    print(json.dumps(brief, indent=1, default=str))
    print(f"[first_contact] report -> {out}")
    return 1 if crashed else (2 if blocking else 0)


if __name__ == "__main__":
    sys.exit(main())
