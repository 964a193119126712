"""Generate golden fixtures from the reference checkout (run in the build container only).

    python tests/golden/make_golden.py [/root/reference]

The reference cannot travel to the GPU box, so what it computes is committed here as small
JSON fixtures:
  smart_resize.json  (h, w, min_pixels, max_pixels) -> (h_bar, w_bar) from the reference's
                     dots_ocr/utils/image_utils.py:29-63, incl. the SURVEY §8(c) known answers,
                     every fixture image size in the reference tree and a seeded random sweep.
  ../../dots_ocr_amd/data/prompts.json   the 8 task prompts (reference dots_ocr/utils/prompts.py:1-46),
                     which are model inputs and must be byte-identical.
  format_transformer.json  input -> output of get_formula_in_markdown / clean_text / layoutjson2md / fix_streamlit_formulas
                     (reference dots_ocr/utils/format_transformer.py) on hand-written and mutated cases.
  output_cleaner.json      input -> OutputCleaner().clean_model_output(input) (reference dots_ocr/utils/output_cleaner.py)
                     on valid, truncated, glued, repeated and degenerate layout JSON, plus a seeded mutation sweep.
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
    post_processing_goldens(rng)


def post_processing_goldens(rng):
    import contextlib
    import io
    ft = _load("dots_ocr.utils.format_transformer", REF / "dots_ocr/utils/format_transformer.py")
    oc = _load("dots_ocr.utils.output_cleaner", REF / "dots_ocr/utils/output_cleaner.py")
    formulas = ["$x^2$", "E = mc", "$$a$$ and $$b$$", "$$ a + b $$", "$$a$b$$", "$$", "$$$", "\\[ x \\]", "see \\[ x \\] here", "\\frac{a}{b}",
                "  \\alpha + \\beta  ", "`\\sum_i x_i`", "\\documentclass{article}\\usepackage{amsmath}\\begin{document}\\frac12\\end{document}",
                "\\usepackage[utf8]{inputenc} \\sqrt{2}", "\\( y \\)", "\\begin{align} a &= b \\end{align}", "plain text", "", "   ", "$", "a $ b", "x\n\\[ y \\]\nz",
                "\\[ a \\] trailing", "$$\na\n$$", "`$x$`", "\\mathbb{R}", "100% \\$5", "\\[\n a \n\\]", "$$ $$"]
    texts = ["`$x$`", " padded ", "", None, "`$a$` and more", "`$`", "`$$`", "plain", "`$x$", "\t`$y$`\n"]
    md_cases = ["$$a$$", "$$\na\n$$", "x $$a$$ y $$\nb$$ z", "no formula", "$$\n\na\n\n$$", "$$a\n$$ $$\nb$$"]
    cells_cases = [
        [{"bbox": [0, 0, 10, 10], "category": "Text", "text": " hello "}, {"bbox": [1.0, 2.0, 3.9, 4.2], "category": "Formula", "text": "\\frac{a}{b}"},
         {"bbox": [0, 0, 5, 5], "category": "Page-header", "text": "head"}, {"bbox": [0, 0, 5, 5], "category": "Page-footer", "text": "foot"},
         {"bbox": [0, 0, 5, 5], "category": "Table", "text": "<table></table>"}, {"bbox": [0, 0, 5, 5], "category": "Title"}],
        [], [{"bbox": ["3", "4", "5", "6"], "category": "Formula", "text": "$x$"}]]
    out = {"get_formula_in_markdown": [[t, ft.get_formula_in_markdown(t)] for t in formulas],
           "clean_text": [[t, ft.clean_text(t)] for t in texts],
           "fix_streamlit_formulas": [[t, ft.fix_streamlit_formulas(t)] for t in md_cases],
           "layoutjson2md": [[c, hf, ft.layoutjson2md(None, c, "text", hf)] for c in cells_cases for hf in (False, True)]}
    (HERE / "format_transformer.json").write_text(json.dumps(out, ensure_ascii=False, indent=1))

    def cell(i, text=None, cat="Text"):
        d = {"bbox": [10 * i, 20 * i, 10 * i + 50, 20 * i + 30], "category": cat}
        if text is not None:
            d["text"] = text
        return d
    valid = json.dumps([cell(1, "alpha"), cell(2, "beta {braces}"), cell(3, None, "Picture"), cell(4, 'delta "quoted"')], ensure_ascii=False)
    cases = [valid, valid[:-1], valid[:-20], valid[:60], valid.replace("}, {", "}{"), valid.replace("}, {", "} {"), valid[1:], valid + ",",
             "[" + ", ".join([json.dumps(cell(1, "rep"))] * 4) + "]", "[" + ", ".join([json.dumps(cell(1, "rep"))] * 4) + ", " + json.dumps(cell(2, "x"))[:25],
             json.dumps([cell(i, "same") for i in range(7)]), json.dumps([cell(i % 2, f"t{i}") for i in range(6)]),
             '[{"bbox": [1, 2, 3, 4], "category": "Text", "text": "unterminated', '[{"bbox": [1, 2, 3], "category": "Text", "text": "unterminated',
             '[{"bbox": [1, 2, 3, 4], "text": "no category', '[{"bbox": [1, 2, x, 4], "category": "Text"', "", "not json at all", "[]", "{}", "42", "null",
             '{"bbox": [1,2,3,4], "category": "Text", "text": "bare object"}', '[{"category": "Text", "text": "no bbox"}]',
             json.dumps([cell(1, "a")]) * 2, valid[:-1] * 30, "[" + ", ".join(json.dumps(cell(i, "long " * 40)) for i in range(400)) + "]",
             [cell(1, "list"), {"bbox": [1, 2, 3], "category": "Text", "text": "three"}, {"bbox": [1, 2, 3]}, {"bbox": "bad", "category": "x"}, "str", {"category": "Title"}, {"text": "orphan"}],
             [cell(1, "dup"), cell(1, "dup2"), cell(2, "p"), cell(3, "p"), cell(4, "p"), cell(5, "p"), cell(6, "p")], [], [cell(1, "only")]]
    for _ in range(120):                                                         # seeded mutation sweep over a valid page
        n = rng.randint(1, 9)
        s = json.dumps([cell(rng.randint(1, 5), rng.choice(["a", "b}", "{c", "d\n", None]), rng.choice(["Text", "Table", "Formula"])) for _ in range(n)])
        for _ in range(rng.randint(1, 3)):
            op = rng.randint(0, 4)
            if op == 0:
                s = s[:rng.randint(0, len(s))]
            elif op == 1:
                s = s.replace("}, {", rng.choice(["}{", "} {", "}\n{"]), rng.randint(1, 3))
            elif op == 2 and len(s) > 4:
                a = rng.randint(0, len(s) - 2)
                s = s[:a] + s[a + rng.randint(1, 3):]
            elif op == 3:
                s = s + s[rng.randint(0, len(s)):]
            else:
                s = s.replace('"bbox"', '"bbox" ', 1)
        cases.append(s)
    res = []
    for c in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            got = oc.OutputCleaner().clean_model_output(c)
        res.append([c, got])
    (HERE / "output_cleaner.json").write_text(json.dumps(res, ensure_ascii=False))
    print(f"wrote {sum(len(v) for v in out.values())} format_transformer cases, {len(res)} output_cleaner cases")


if __name__ == "__main__":
    main()
