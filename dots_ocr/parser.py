"""`dots_ocr.parser` (reference dots_ocr/parser.py) served by the MI355X engine."""
from dots_ocr_amd.parser import DotsOCRParser, main  # noqa: F401

if __name__ == "__main__":
    main()
