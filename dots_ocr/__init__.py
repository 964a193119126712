"""Drop-in for the reference's `dots_ocr` package (dots_ocr/__init__.py:1): same import paths,
implemented by the MI355X-native engine in `dots_ocr_amd`."""
from dots_ocr_amd.parser import DotsOCRParser  # noqa: F401
