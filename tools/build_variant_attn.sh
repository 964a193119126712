#!/bin/bash
# tools/build_variant_attn.sh NAME "EXTRA_FLAGS": rebuilds attn_prefill.hip with extra -D flags into tools/bin/var_NAME/libdots_ocr_hip.so
set -e
cd "$(dirname "$0")/.."
d=tools/bin/var_$1; mkdir -p $d
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Idots_ocr_amd/csrc -Iinclude $2"
hipcc $FLAGS -x hip -c dots_ocr_amd/csrc/attn_prefill.hip -o $d/attn_prefill.hip.o 2>/dev/null
objs=$(ls dots_ocr_amd/_obj/*.o | grep -v "/attn_prefill.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $objs $d/attn_prefill.hip.o -o $d/libdots_ocr_hip.so
echo built $d
