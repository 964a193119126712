"""Salvage of malformed layout JSON (reference dots_ocr/utils/output_cleaner.py, called from layout_utils.py:221-228 when
the generation does not parse — the usual case for long pages that stop at max_new_tokens).  Host-side post-processing of
the hot path's OUTPUT string, outside the accelerated path; restated as a small pipeline of pure
functions and pinned input-by-input against the reference class by tests/golden/output_cleaner.json."""
from __future__ import annotations

import json
import re
from typing import Any, List, Optional

_CELL = re.compile(r'\{[^{}]*?"bbox"\s*:\s*\[[^\]]*?\][^{}]*?\}', re.DOTALL)     # one flat {..."bbox": [...]...} object
_GLUED = re.compile(r'\}\s*\{(?!")')                                               # "}{": a lost "," between two cells
_OPEN = '{"bbox":'
LONG_TEXT = 50000


def _clean_list(items: list) -> list:
    """Parsed but irregular cells: a 3-number bbox is dropped (category/text kept), other malformed boxes drop the cell."""
    out = []
    for it in items:
        if not isinstance(it, dict):
            continue
        if "bbox" not in it:
            if "category" in it:
                out.append(dict(it))
            continue
        box = it["bbox"]
        if isinstance(box, list) and len(box) == 4:
            out.append(dict(it))
        elif isinstance(box, list) and len(box) == 3:
            kept = {k: it[k] for k in ("category", "text") if k in it}
            if kept:
                out.append(kept)
    return out


def _drop_unfinished_tail(text: str) -> str:
    """A generation cut off mid-cell (or suspiciously long): everything from the last cell opener on is discarded."""
    if len(text) <= LONG_TEXT and text.strip().endswith("]"):
        return text
    if text.count(_OPEN) <= 1:
        return text
    at = text.rfind(_OPEN)
    if at <= 0:
        return text
    head = text[:at].rstrip()
    return head[:-1] if head.endswith(",") else head


def _dedupe_cells(text: str) -> str:
    """Repetition loops: identical cell objects are kept once, in order (the text is rebuilt only if something repeated)."""
    cells = [m.group() for m in _CELL.finditer(text)]
    uniq = list(dict.fromkeys(cells))
    return text if len(uniq) == len(cells) else "[" + ", ".join(uniq) + "]"


def _as_array(text: str) -> str:
    text = text.strip()
    if not text.startswith("["):
        text = "[" + text
    if not text.endswith("]"):
        text = text.rstrip(",").rstrip() + "]"
    return text


def _lone_unfinished_cell(text: str) -> Optional[list]:
    """'[{"bbox": [a,b,c,d], "category": .., "text": "...' with nothing closed: rebuild that one cell."""
    if not text.strip().startswith('[{"bbox":'):
        return None
    try:
        m = re.search(r'"bbox"\s*:\s*\[([^\]]+)\]', text)
        if not m:
            return None
        box = [int(x.strip()) for x in m.group(1).split(",")]
        if len(box) != 4:
            return None
        cat = re.search(r'"category"\s*:\s*"([^"]+)"', text)
        cell = {"bbox": box, "category": cat.group(1) if cat else "Text"}
        body = re.search(r'"text"\s*:\s*"([^"]{0,10000})', text)
        if body and body.group(1):
            cell["text"] = body.group(1)
        return [cell]
    except Exception:
        return None


def _parse_array(text: str) -> Optional[list]:
    try:
        data = json.loads(text)
        return data if isinstance(data, list) else None
    except json.JSONDecodeError:
        pass
    good = []
    for m in _CELL.finditer(text):
        try:
            good.append(json.loads(m.group()))
        except Exception:
            continue
    return good or _lone_unfinished_cell(text)


def _clean_string(text: str) -> Optional[list]:
    text = _GLUED.sub("},{", text)
    text = _as_array(_dedupe_cells(_drop_unfinished_tail(text)))
    return _parse_array(text)


def _drop_repeats(cells: list) -> list:
    """Degenerate repetition inside a parsed page: a (category, text) pair seen >= 5 times or a bbox seen >= 2 times keeps
    only its first occurrence."""
    if len(cells) <= 1:
        return cells
    pairs, boxes = {}, {}
    for i, c in enumerate(cells):
        if not isinstance(c, dict):
            continue
        if "category" in c and "text" in c:
            pairs.setdefault((c.get("category", ""), c.get("text", "")), []).append(i)
        box = c.get("bbox")
        if "bbox" in c and isinstance(box, list) and len(box) > 0:
            boxes.setdefault(tuple(box), []).append(i)
    drop = set()
    for where in pairs.values():
        if len(where) >= 5:
            drop.update(where[1:])
    for where in boxes.values():
        if len(where) >= 2:
            drop.update(where[1:])
    return [c for i, c in enumerate(cells) if i not in drop] if drop else cells


class OutputCleaner:
    """Same entry point as the reference class: clean_model_output(str | list) -> list of cells ([] when nothing could be
    recovered); an unexpected error returns the input unchanged."""

    def clean_model_output(self, model_output: Any):
        try:
            cells = _clean_list(model_output) if isinstance(model_output, list) else _clean_string(str(model_output))
            if cells is None:
                return []
            return _drop_repeats(cells) if cells else cells
        except Exception:
            return model_output
