"""Input-contract constants (reference dots_ocr/utils/consts.py:1-5)."""
MIN_PIXELS = 3136
MAX_PIXELS = 11289600
IMAGE_FACTOR = 28
image_extensions = {'.jpg', '.jpeg', '.png'}
