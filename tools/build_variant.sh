#!/bin/bash
# tools/build_variant.sh NAME "EXTRA_FLAGS": rebuilds the decode kernels with extra -D flags into tools/bin/var_NAME/libdots_ocr_hip.so
# (the other objects come from dots_ocr_amd/_obj), for A/B runs of tools/bin/decode_bench via LD_LIBRARY_PATH.
set -e
cd "$(dirname "$0")/.."
d=tools/bin/var_$1; mkdir -p $d
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Idots_ocr_amd/csrc -Iinclude $2"
for f in decode.hip decode_fused.hip; do hipcc $FLAGS -x hip -c dots_ocr_amd/csrc/$f -o $d/$f.o 2>/dev/null & done; wait
objs=$(ls dots_ocr_amd/_obj/*.o | grep -v "/decode.hip.o\|/decode_fused.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $objs $d/decode.hip.o $d/decode_fused.hip.o -o $d/libdots_ocr_hip.so
echo built $d
