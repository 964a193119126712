from dots_ocr_amd.consts import *  # noqa: F401,F403
