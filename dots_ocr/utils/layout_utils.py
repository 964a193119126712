from dots_ocr_amd.layout_utils import *  # noqa: F401,F403
