"""Image helpers the reference's demo UIs import (reference dots_ocr/utils/demo_utils/display.py:5-62): same names and results."""
from pathlib import Path

from PIL import Image

_IMAGE_SUFFIXES = {".jpg", ".jpeg", ".png", ".gif", ".bmp"}


def is_valid_image_path(image_path) -> bool:
    p = Path(image_path)
    return p.exists() and p.suffix.lower() in _IMAGE_SUFFIXES


def read_image(image_path, use_native: bool = False):
    """-> (image resized so that its longer side is 1024 px, or left at its own longer side with use_native; width; height)."""
    if not is_valid_image_path(image_path):
        raise FileNotFoundError(f"{image_path}: Image path does not exist")
    image = Image.open(image_path)
    w, h = image.size
    longest = max(w, h) if use_native else 1024
    size = (longest, int(h * longest / w)) if w > h else (int(w * longest / h), longest)
    return image.resize(size), w, h
