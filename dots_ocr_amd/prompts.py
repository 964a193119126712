"""Task prompts keyed by mode (reference dots_ocr/utils/prompts.py:1-46).

The strings are model inputs and must be byte-identical to the reference's, so they are kept as
data (data/prompts.json, generated from the reference by tests/golden/make_golden.py and checked
against it in tests/test_host_cpu.py) rather than retyped here.
"""
import json
from pathlib import Path

dict_promptmode_to_prompt = json.loads((Path(__file__).resolve().parent / "data" / "prompts.json").read_text(encoding="utf-8"))
