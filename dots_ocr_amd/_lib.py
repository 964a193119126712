"""ctypes loader for the in-tree HIP engine (libdots_ocr_hip.so).

There is NO CPU fallback: if the library is missing or cannot be loaded the import of any
product module fails loudly (the oracle under /oracle is test infrastructure only and is
never imported from here).
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

_LIB = None


class HipEngineUnavailable(RuntimeError):
    pass


def lib_path() -> Path:
    """The in-tree build; DOTS_OCR_LIB points at another build of the same library (A/B runs of kernel variants)."""
    override = os.environ.get("DOTS_OCR_LIB")
    return Path(override) if override else Path(__file__).resolve().parent / "lib" / "libdots_ocr_hip.so"


def load() -> ctypes.CDLL:
    """Load (once) and return the engine library.  Builds it if DOTS_OCR_AUTOBUILD=1."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not path.exists() and os.environ.get("DOTS_OCR_AUTOBUILD", "0") == "1":
        from . import build as _b
        _b.build()
    if not path.exists():
        raise HipEngineUnavailable(
            f"{path} not found: build it with `python -m dots_ocr_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the dots.ocr HIP engine.")
    try:
        _LIB = ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # missing libamdhip64 etc.
        raise HipEngineUnavailable(f"cannot load {path}: {e}") from e
    return _LIB
