#!/bin/bash
# tools/build_variant_gemm.sh NAME "EXTRA_FLAGS": rebuilds gemm.hip with extra -D flags into tools/bin/var_NAME/libdots_ocr_hip.so
# (the other objects come from dots_ocr_amd/_obj) for A/B runs: DOTS_OCR_LIB=tools/bin/var_NAME/libdots_ocr_hip.so python tools/gemm_bench.py
set -e
cd "$(dirname "$0")/.."
d=tools/bin/var_$1; mkdir -p $d
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Idots_ocr_amd/csrc -Iinclude $2"
hipcc $FLAGS -x hip -c dots_ocr_amd/csrc/gemm.hip -o $d/gemm.hip.o 2>/dev/null
objs=$(ls dots_ocr_amd/_obj/*.o | grep -v "/gemm.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $objs $d/gemm.hip.o -o $d/libdots_ocr_hip.so
echo built $d
