"""Host-side handling of the model's layout JSON (reference dots_ocr/utils/layout_utils.py:115-228).
This consumes the hot path's OUTPUT string and is outside the accelerated path; what
DotsOCRParser needs is provided: bbox rescaling both ways, JSON decoding with a failure flag, and the
reference's salvage of malformed generations (output_cleaner.OutputCleaner, layout_utils.py:221-228)."""
from __future__ import annotations

import json
from typing import Dict, List

from .consts import MAX_PIXELS, MIN_PIXELS
from .image_utils import smart_resize


def _scales(origin_image, input_width, input_height, min_pixels, max_pixels):
    ow, oh = origin_image.size
    ih, iw = smart_resize(input_height, input_width, min_pixels=min_pixels or MIN_PIXELS, max_pixels=max_pixels or MAX_PIXELS)
    return iw / ow, ih / oh


def pre_process_bboxes(origin_image, bboxes, input_width, input_height, factor: int = 28, min_pixels=None, max_pixels=None):
    """original-image boxes -> model-input space (for the grounding prompt)."""
    assert isinstance(bboxes, list) and len(bboxes) > 0 and isinstance(bboxes[0], list)
    sx, sy = _scales(origin_image, input_width, input_height, min_pixels, max_pixels)
    return [[int(float(b[0]) * sx), int(float(b[1]) * sy), int(float(b[2]) * sx), int(float(b[3]) * sy)] for b in bboxes]


def post_process_cells(origin_image, cells: List[Dict], input_width, input_height, factor: int = 28,
                       min_pixels=None, max_pixels=None) -> List[Dict]:
    """model-input-space boxes -> original-image space."""
    assert isinstance(cells, list) and len(cells) > 0 and isinstance(cells[0], dict)
    sx, sy = _scales(origin_image, input_width, input_height, min_pixels, max_pixels)
    out = []
    for cell in cells:
        b = cell["bbox"]
        c = dict(cell)
        c["bbox"] = [int(float(b[0]) / sx), int(float(b[1]) / sy), int(float(b[2]) / sx), int(float(b[3]) / sy)]
        out.append(c)
    return out


def post_process_output(response, prompt_mode, origin_image, input_image, min_pixels=None, max_pixels=None):
    """-> (cells, filtered).  Plain-text modes return the response unchanged."""
    if prompt_mode in ("prompt_ocr", "prompt_table_html", "prompt_table_latex", "prompt_formula_latex"):
        return response
    cells = response
    try:
        cells = json.loads(cells)
        return post_process_cells(origin_image, cells, input_image.width, input_image.height, min_pixels=min_pixels, max_pixels=max_pixels), False
    except Exception as e:                       # malformed generation (typically cut off at max_new_tokens)
        print(f"cells post process error: {e}, when using {prompt_mode}")
    from .output_cleaner import OutputCleaner
    salvaged = OutputCleaner().clean_model_output(cells)      # `cells`: the parsed object if only the rescaling failed
    if isinstance(salvaged, list):
        salvaged = "\n\n".join(c["text"] for c in salvaged if "text" in c)
    return salvaged, True


def is_legal_bbox(cells) -> bool:
    return all(c["bbox"][2] > c["bbox"][0] and c["bbox"][3] > c["bbox"][1] for c in cells)


def draw_layout_on_image(image, cells):
    """Outline every cell on a copy of the page (PIL; the reference draws with fitz)."""
    from PIL import ImageDraw
    img = image.copy()
    d = ImageDraw.Draw(img)
    for cell in cells:
        x1, y1, x2, y2 = cell["bbox"]
        d.rectangle([x1, y1, x2, y2], outline=(255, 0, 0), width=2)
        d.text((x1 + 2, max(0, y1 - 10)), str(cell.get("category", "")), fill=(255, 0, 0))
    return img
