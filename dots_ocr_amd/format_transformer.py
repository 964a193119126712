"""Layout cells -> Markdown (reference dots_ocr/utils/format_transformer.py:145-180).  Post-processing
of the hot path's output, outside the accelerated path; the subset DotsOCRParser calls."""
from __future__ import annotations

from .image_utils import PILimage_to_base64


def _formula(text: str) -> str:
    t = text.strip()
    for a, b in (("$$", "$$"), ("\\[", "\\]"), ("$", "$")):
        if t.startswith(a) and t.endswith(b) and len(t) >= len(a) + len(b):
            t = t[len(a):len(t) - len(b)].strip()
            break
    return f"$$\n{t}\n$$"


def layoutjson2md(image, cells, text_key: str = "text", no_page_hf: bool = False) -> str:
    parts = []
    for cell in cells:
        cat = cell.get("category", "")
        if no_page_hf and cat in ("Page-header", "Page-footer"):
            continue
        if cat == "Picture":
            x1, y1, x2, y2 = cell["bbox"]
            parts.append(f"![]({PILimage_to_base64(image.crop((x1, y1, x2, y2)))})")
        elif cat == "Formula":
            parts.append(_formula(cell.get(text_key, "")))
        else:
            parts.append(str(cell.get(text_key, "")).strip())
    return "\n\n".join(parts)
