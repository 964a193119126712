"""Host-side image handling of the dots.ocr input contract.

Mirrors the reference's dots_ocr/utils/image_utils.py (smart_resize :29-63, to_rgb :74-80,
fetch_image :84-140, PILimage_to_base64 :67-71) and the Qwen2-VL image processor the checkpoint
selects (resize -> 1/255 -> normalise -> patchify; transformers
models/qwen2_vl/image_processing_pil_qwen2_vl.py:126-246) with the same names, argument meaning
and error behaviour, so that image size -> number of patches is identical to the reference.
PDF rasterisation (fitz) is imported lazily: it is outside the accelerated path.
"""
from __future__ import annotations

import base64
import copy
import functools
import math
from io import BytesIO
from typing import Sequence, Tuple

import numpy as np
from PIL import Image

from .consts import IMAGE_FACTOR, MAX_PIXELS, MIN_PIXELS


def round_by_factor(number: float, factor: int) -> int:
    return round(number / factor) * factor


def ceil_by_factor(number: float, factor: int) -> int:
    return math.ceil(number / factor) * factor


def floor_by_factor(number: float, factor: int) -> int:
    return math.floor(number / factor) * factor


def smart_resize(height: int, width: int, factor: int = IMAGE_FACTOR, min_pixels: int = MIN_PIXELS,
                 max_pixels: int = MAX_PIXELS) -> Tuple[int, int]:
    """(h, w) -> (h_bar, w_bar): both multiples of `factor`, area clamped to [min_pixels, max_pixels]
    with the aspect ratio kept; aspect ratios above 200 are rejected."""
    ratio = max(height, width) / min(height, width)
    if ratio > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {ratio}")
    h_bar = max(factor, round_by_factor(height, factor))
    w_bar = max(factor, round_by_factor(width, factor))
    area = h_bar * w_bar
    if area > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, floor_by_factor(height / beta, factor))
        w_bar = max(factor, floor_by_factor(width / beta, factor))
    elif area < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar, w_bar = ceil_by_factor(height * beta, factor), ceil_by_factor(width * beta, factor)
        if h_bar * w_bar > max_pixels:       # the token budget wins over the minimum size
            beta = math.sqrt((h_bar * w_bar) / max_pixels)
            h_bar = max(factor, floor_by_factor(h_bar / beta, factor))
            w_bar = max(factor, floor_by_factor(w_bar / beta, factor))
    return h_bar, w_bar


def PILimage_to_base64(image: Image.Image, format: str = "PNG") -> str:
    buf = BytesIO()
    image.save(buf, format=format)
    return f"data:image/{format.lower()};base64," + base64.b64encode(buf.getvalue()).decode("utf-8")


def to_rgb(pil_image: Image.Image) -> Image.Image:
    if pil_image.mode != "RGBA":
        return pil_image.convert("RGB")
    canvas = Image.new("RGB", pil_image.size, (255, 255, 255))
    canvas.paste(pil_image, mask=pil_image.split()[3])
    return canvas


def _open(image) -> Image.Image:
    if isinstance(image, Image.Image):
        return image
    if not isinstance(image, str):
        raise ValueError(f"Unrecognized image input, support local path, http url, base64 and PIL.Image, got {image}")
    if image.startswith(("http://", "https://")):
        import requests
        with requests.get(image, stream=True) as resp:
            resp.raise_for_status()
            with BytesIO(resp.content) as bio:
                return copy.deepcopy(Image.open(bio))
    if image.startswith("file://"):
        return Image.open(image[7:])
    if image.startswith("data:image"):
        if "base64," not in image:
            raise ValueError(f"Unrecognized image input, support local path, http url, base64 and PIL.Image, got {image}")
        with BytesIO(base64.b64decode(image.split("base64,", 1)[1])) as bio:
            return copy.deepcopy(Image.open(bio))
    return Image.open(image)


def fetch_image(image, min_pixels=None, max_pixels=None, resized_height=None, resized_width=None) -> Image.Image:
    """path / URL / data-URL / PIL -> RGB PIL image; resized only when bounds (or an explicit size) are given."""
    assert image is not None, f"image not found, maybe input format error: {image}"
    img = to_rgb(_open(image))
    if resized_height and resized_width:
        rh, rw = smart_resize(resized_height, resized_width, factor=IMAGE_FACTOR)
    elif min_pixels or max_pixels:
        w, h = img.size
        rh, rw = smart_resize(h, w, factor=IMAGE_FACTOR, min_pixels=min_pixels or MIN_PIXELS, max_pixels=max_pixels or MAX_PIXELS)
    else:
        return img
    assert rh > 0 and rw > 0, f"resized_height: {rh}, resized_width: {rw}, min_pixels: {min_pixels}, max_pixels:{max_pixels}"
    return img.resize((rw, rh))


def get_input_dimensions(image, min_pixels: int, max_pixels: int, factor: int = 28):
    """(width, height) the model sees for `image` (reference image_utils.py:142-167; used by the demo UIs)."""
    h, w = smart_resize(image.height, image.width, factor=factor, min_pixels=min_pixels, max_pixels=max_pixels)
    return w, h


def get_image_by_fitz_doc(image, target_dpi: int = 200):
    """Re-render an image at `target_dpi` through a PDF round trip (needs PyMuPDF; reference :170-196)."""
    import fitz  # noqa: F401  (optional dependency, outside the accelerated path)
    from .doc_utils import fitz_doc_to_image
    img = _open(image)
    buf = BytesIO()
    img.save(buf, format=img.format or "PNG")
    src = fitz.open(stream=buf.getvalue(), filetype="png" if (img.format or "PNG").upper() == "PNG" else "jpeg")
    pdf = fitz.open("pdf", src.convert_to_pdf())
    return fitz_doc_to_image(pdf[0], target_dpi=target_dpi)


# ------------------------------------------------------------------------------ image processor
def preprocess_image(image: Image.Image, patch_size: int = 14, merge_size: int = 2, temporal_patch_size: int = 1,
                     min_pixels: int = MIN_PIXELS, max_pixels: int = MAX_PIXELS,
                     image_mean: Sequence[float] = (0.48145466, 0.4578275, 0.40821073),
                     image_std: Sequence[float] = (0.26862954, 0.26130258, 0.27577711)):
    """One RGB image -> (pixel_values float32 [gh*gw, C*T*P*P], [t, gh, gw]).
    Patches are emitted block-major over merge x merge groups, channel-major inside a patch."""
    img = to_rgb(image)
    w, h = img.size
    rh, rw = smart_resize(h, w, patch_size * merge_size, min_pixels, max_pixels)
    if (rh, rw) != (h, w):
        img = img.resize((rw, rh), resample=Image.BICUBIC)
    px = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) * np.float32(1.0 / 255.0)
    px = (px - np.asarray(image_mean, np.float32).reshape(-1, 1, 1)) / np.asarray(image_std, np.float32).reshape(-1, 1, 1)
    c = px.shape[0]
    gh, gw = rh // patch_size, rw // patch_size
    m = merge_size
    tiles = px.reshape(c, gh // m, m, patch_size, gw // m, m, patch_size).transpose(1, 4, 2, 5, 0, 3, 6)
    if temporal_patch_size > 1:
        tiles = np.repeat(tiles[:, :, :, :, :, None], temporal_patch_size, axis=5)
    flat = np.ascontiguousarray(tiles.reshape(gh * gw, c * temporal_patch_size * patch_size * patch_size), dtype=np.float32)
    return flat, [1, gh, gw]


# ------------------------------------------------------------------------------ device preprocessing (host half)
@functools.lru_cache(maxsize=64)
def bicubic_resample_tables(in_size: int, out_size: int):
    """Fixed-point tap tables of Pillow's BICUBIC resampler for one axis (vectorised): int32 coeffs [out, ksize] and
    int32 bounds [out, 2] = (first input index, tap count).  The GPU kernels (csrc/preprocess.hip) apply them with
    Pillow's exact integer arithmetic (22 fractional bits, round, clip to uint8 after each pass), so the device path
    is bit-identical to `Image.resize(..., BICUBIC)` used by `preprocess_image`."""
    bits = 22
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast truncates; values are >= -support > INT_MIN
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    x = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale))
    a = -0.5
    w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
    valid = taps < xmax[:, None]
    w = np.where(valid, w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):                                                     # same left-to-right summation order as the C loop
        ww = np.where(valid[:, j], ww + w[:, j], ww)
    wn = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(wn < 0, (-0.5 + wn * (1 << bits)), (0.5 + wn * (1 << bits)))
    coeffs = np.where(valid, np.trunc(fixed), 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return np.ascontiguousarray(coeffs), np.ascontiguousarray(bounds)
