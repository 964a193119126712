from dots_ocr_amd.doc_utils import *  # noqa: F401,F403
