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


# category -> RGB of the overlay (the reference's table, layout_utils.py:13-27; its 4th component is unused there as well)
dict_layout_type_to_color = {
    "Text": (0, 128, 0), "Picture": (255, 0, 255), "Caption": (255, 165, 0), "Section-header": (0, 255, 255),
    "Footnote": (0, 128, 0), "Formula": (128, 128, 128), "Table": (255, 192, 203), "Title": (255, 0, 0),
    "List-item": (0, 0, 255), "Page-header": (0, 128, 0), "Page-footer": (128, 0, 128), "Other": (165, 42, 42),
    "Unknown": (0, 0, 0),
}


def draw_layout_on_image(image, cells, resized_height=None, resized_width=None, fill_bbox=True, draw_bbox=True):
    """The page with every cell drawn on it in its category colour and labelled "<reading order>_<category>" to the right of its
    top edge (reference layout_utils.py:30-110, which renders through a PDF page with fitz; here PIL, so the result is the same
    picture up to font and anti-aliasing).  fill_bbox: translucent fill (30 %) instead of an outline; draw_bbox=False: labels only;
    resized_height / resized_width: the boxes are in the coordinates of a page resized to that size and are scaled back."""
    from PIL import Image, ImageDraw, ImageFont
    base = image.convert("RGBA")
    overlay = Image.new("RGBA", base.size, (0, 0, 0, 0))
    d = ImageDraw.Draw(overlay)
    sx = sy = 1.0
    if resized_height and resized_width:
        sx, sy = resized_width / base.width, resized_height / base.height
    try:
        font = ImageFont.load_default(size=20)
    except TypeError:                                  # Pillow < 10.1: only the small bitmap font
        font = ImageFont.load_default()
    for order, cell in enumerate(cells):
        x0, y0, x1, y1 = cell["bbox"]
        if sx != 1.0 or sy != 1.0:
            x0, y0, x1, y1 = int(x0 / sx), int(y0 / sy), int(x1 / sx), int(y1 / sy)
        category = cell["category"]
        rgb = dict_layout_type_to_color.get(category, (0, 128, 0))
        if draw_bbox:
            if fill_bbox:
                d.rectangle([x0, y0, x1, y1], fill=rgb + (77,))            # 0.3 opacity
            else:
                d.rectangle([x0, y0, x1, y1], outline=rgb + (255,), width=1)
        d.text((x1, y0), f"{order}_{category}", fill=rgb + (255,), font=font)
    return Image.alpha_composite(base, overlay).convert("RGB")
