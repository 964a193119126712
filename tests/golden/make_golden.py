"""Generate golden fixtures from the reference checkout (run in the build container only).

    python tests/golden/make_golden.py [/root/reference]

The reference cannot travel to the GPU box, so what it computes is committed here as small
JSON fixtures:
  smart_resize.json  (h, w, min_pixels, max_pixels) -> (h_bar, w_bar) from the reference's
                     dots_ocr/utils/image_utils.py:29-63, incl. the SURVEY §8(c) known answers,
                     every fixture image size in the reference tree and a seeded random sweep.
  ../../dots_ocr_amd/data/prompts.json   the 8 task prompts (reference dots_ocr/utils/prompts.py:1-46),
                     which are model inputs and must be byte-identical.
"""
import importlib.util
import json
import random
import sys
import types
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
HERE = Path(__file__).resolve().parent


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    # stub the packages the reference imports but this container lacks (fitz = PyMuPDF)
    sys.modules.setdefault("fitz", types.ModuleType("fitz"))
    pkg = types.ModuleType("dots_ocr"); pkg.__path__ = [str(REF / "dots_ocr")]
    utils = types.ModuleType("dots_ocr.utils"); utils.__path__ = [str(REF / "dots_ocr" / "utils")]
    sys.modules["dots_ocr"] = pkg
    sys.modules["dots_ocr.utils"] = utils
    _load("dots_ocr.utils.consts", REF / "dots_ocr/utils/consts.py")
    du = types.ModuleType("dots_ocr.utils.doc_utils"); du.fitz_doc_to_image = None
    sys.modules["dots_ocr.utils.doc_utils"] = du
    iu = _load("dots_ocr.utils.image_utils", REF / "dots_ocr/utils/image_utils.py")
    pr = _load("dots_ocr.utils.prompts", REF / "dots_ocr/utils/prompts.py")

    cases = [(2250, 1700), (2339, 1654), (1344, 1344), (4500, 4500), (20, 20), (3360, 3360),
             (550, 583), (1024, 946), (3308, 2339), (28, 28), (27, 5000), (1, 199), (56, 56),
             (14, 14), (10000, 10000), (3000, 4000), (4000, 3000), (100, 20000), (3361, 3359)]
    from PIL import Image
    for p in sorted(list((REF / "demo").glob("*.jpg")) + list((REF / "demo").glob("*.png"))
                    + list((REF / "assets/showcase/origin").glob("*"))):
        try:
            with Image.open(p) as im:
                cases.append((im.height, im.width))
        except Exception:
            pass
    rng = random.Random(20260921)
    for _ in range(300):
        cases.append((rng.randint(1, 6000), rng.randint(1, 6000)))
    out = []
    bounds = [(3136, 11289600), (3136, 1003520), (200000, 11289600), (3136, 3136 * 4)]
    for (h, w) in cases:
        for (mn, mx) in bounds[: 1 if len(out) > 400 else 4]:
            try:
                r = list(iu.smart_resize(h, w, 28, mn, mx))
            except ValueError:
                r = "ValueError"
            out.append({"h": h, "w": w, "min_pixels": mn, "max_pixels": mx, "out": r})
    (HERE / "smart_resize.json").write_text(json.dumps(out))
    data = HERE.parent.parent / "dots_ocr_amd" / "data"
    data.mkdir(exist_ok=True)
    (data / "prompts.json").write_text(json.dumps(pr.dict_promptmode_to_prompt, ensure_ascii=False, indent=1))
    (HERE / "prompts.json").write_text(json.dumps(pr.dict_promptmode_to_prompt, ensure_ascii=False, indent=1))
    print(f"wrote {len(out)} smart_resize cases, {len(pr.dict_promptmode_to_prompt)} prompts")


if __name__ == "__main__":
    main()
