from dots_ocr_amd.output_cleaner import *  # noqa: F401,F403
from dots_ocr_amd.output_cleaner import OutputCleaner  # noqa: F401
