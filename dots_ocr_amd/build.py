"""Build the gfx950 HIP engine into an in-tree shared library.

hipcc cross-compiles without a GPU, so this runs both in the CPU build container and on
the MI355X box.  Output: dots_ocr_amd/lib/libdots_ocr_hip.so (git-ignored, travels with
the gpurun snapshot).  Each translation unit is compiled to an object in parallel and
re-used when its source (and every header) is older than the object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
OBJ = PKG / "_obj"
LIBDIR = PKG / "lib"
LIBNAME = "libdots_ocr_hip.so"
ARCH = "gfx950"

CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result",
    f"-I{INCLUDE}", f"-I{CSRC}",
]
LDFLAGS = ["-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the dots.ocr HIP engine cannot be built")
    return exe


def _sources() -> list[Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _headers_mtime() -> float:
    hs = list(CSRC.glob("*.h")) + list(CSRC.glob("*.hpp")) + list(INCLUDE.glob("*.h"))
    return max([h.stat().st_mtime for h in hs], default=0.0)


def lib_path() -> Path:
    return LIBDIR / LIBNAME


def build(force: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    cc = hipcc()
    hdr_m = max(_headers_mtime(), Path(__file__).stat().st_mtime)
    jobs = []
    objs = []
    for src in _sources():
        obj = OBJ / (src.name + ".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_m):
            cmd = [cc, *CXXFLAGS, "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", str(src), "-o", str(obj)]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        p = subprocess.run(cmd, capture_output=True, text=True)
        return src, p

    if jobs:
        if verbose:
            print(f"[dots_ocr_amd.build] compiling {len(jobs)} file(s) for {ARCH}", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, p in ex.map(run, jobs):
                if p.returncode != 0:
                    raise RuntimeError(f"hipcc failed on {src.name}:\n{p.stdout}\n{p.stderr}")
                # per-kernel register / LDS / scratch report (tests/test_cabi_cpu.py asserts that nothing spills)
                usage = [ln for ln in p.stderr.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" in ln]
                (OBJ / (src.name + ".resusage.txt")).write_text("\n".join(usage) + "\n")
                rest = [ln for ln in p.stderr.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" not in ln
                        and "remark" not in ln and ln.strip() and not ln.startswith(" ") and "generated" not in ln]
                if verbose and rest:
                    print("\n".join(rest), file=sys.stderr)
    out = lib_path()
    if jobs or not out.exists():
        cmd = [cc, f"--offload-arch={ARCH}", *LDFLAGS, *map(str, objs), "-o", str(out)]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"link failed:\n{p.stdout}\n{p.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
