from dots_ocr_amd.format_transformer import *  # noqa: F401,F403
