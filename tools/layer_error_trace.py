#!/usr/bin/env python3
"""Where does the engine's logit error enter?  (VERDICT r1 weak #3 / next #7.)  Run on the GPU box:

    python tools/layer_error_trace.py [--tiny] [--out profiles/r02_layer_error_trace.json]

Full dots.ocr architecture (or --tiny), seeded random weights, one 583x550 page.  The engine keeps the bf16 residual stream
after every ViT block and every LM prefill layer (dots_debug_capture_hidden); the CPU oracle returns the same tensors in its
bf16-emulated and fp32 modes (return_hidden=True).  Per layer: max / mean |engine - oracle| relative to the tensor's RMS,
against both modes — the layer where the error vs the EMULATED oracle grows is where a rounding point differs; error that
only grows vs fp32 is bf16 storage noise that the emulation reproduces."""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r02_layer_error_trace.json"))
    a = ap.parse_args()
    from dots_ocr_amd.config import DotsConfig
    from dots_ocr_amd.engine import Engine
    from dots_ocr_amd.image_utils import preprocess_image
    from dots_ocr_amd.synthetic import synth_page, synth_prompt_ids
    from dots_ocr_amd.weights import random_state_dict
    from oracle import model as om
    cfg = DotsConfig.tiny(layers=3, v_layers=3) if a.tiny else DotsConfig()
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    sd = random_state_dict(cfg, seed=0, threads=min(32, os.cpu_count() or 8))
    pv, thw = preprocess_image(synth_page(3, (583, 550)))
    ids = synth_prompt_ids(cfg, pv.shape[0] // 4, n_text_tokens=64, seed=3)
    N, T, E, H = pv.shape[0], len(ids), cfg.vision.embed_dim, cfg.hidden_size
    eng = Engine(cfg, max_batch=1, max_seq_len=T + 64, max_patches=N + 64, max_prefill_tokens=T + 64)
    eng.load_state_dict(sd)
    eng.capture_hidden(cfg.vision.num_hidden_layers * N * E + cfg.num_hidden_layers * T * H)
    eng.vit_forward(pv, np.asarray([thw], np.int64))
    eng.prefill(ids, np.asarray([T], np.int32))
    logits = eng.get_logits()[0].copy()

    def bf(u16):
        return torch.from_numpy((u16.astype(np.uint32) << 16).view(np.float32))
    vit_h = [bf(eng.read_hidden("vit", i)) for i in range(cfg.vision.num_hidden_layers)]
    lm_h = [bf(eng.read_hidden("lm", i)) for i in range(cfg.num_hidden_layers)]
    eng.close()

    sd32 = {k: v.float() for k, v in sd.items()}
    rep = {"model": "tiny" if a.tiny else "dots.ocr architecture, seeded random weights", "input": f"{N} patches, {T} prompt tokens", "vit": [], "lm": []}
    refs = {}
    for name, emu in (("emulated_bf16", True), ("fp32", False)):
        vis, vh = om.vision_tower(sd32, cfg, torch.from_numpy(pv), torch.tensor([thw]), emulate_bf16=emu, return_hidden=True)
        emb = om.build_embeds(sd32, cfg, torch.from_numpy(ids.astype(np.int64)), vis)
        if emu:
            emb = om._r(emb, True)
        lg, lh = om.lm_forward(sd32, cfg, emb, om.KVCache(cfg.num_hidden_layers), emu, return_hidden=True)
        refs[name] = (vh, lh, lg[0])
    for key, got in (("vit", vit_h), ("lm", lm_h)):
        for i, g in enumerate(got):
            row = {"layer": i}
            for name in refs:
                r = refs[name][0 if key == "vit" else 1][i]
                d = (g - r).abs()
                rms = float(r.pow(2).mean().sqrt())
                row[f"max_rel_err_vs_{name}"] = float(d.max()) / rms
                row[f"mean_rel_err_vs_{name}"] = float(d.mean()) / rms
            rep[key].append(row)
    e = torch.from_numpy(logits)
    rep["logits"] = {f"max_abs_err_vs_{n}": float((e - refs[n][2]).abs().max()) for n in refs}
    rep["logits"]["logit_std"] = float(refs["fp32"][2].std())
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(rep, indent=1))
    for key in ("vit", "lm"):
        for row in rep[key][:: max(1, len(rep[key]) // 8)] + rep[key][-1:]:
            print(key, row)
    print(rep["logits"])


if __name__ == "__main__":
    main()
