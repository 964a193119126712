from dots_ocr_amd.image_utils import *  # noqa: F401,F403
from dots_ocr_amd.image_utils import smart_resize, fetch_image, to_rgb, PILimage_to_base64, get_image_by_fitz_doc, round_by_factor, ceil_by_factor, floor_by_factor  # noqa: F401
