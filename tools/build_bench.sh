#!/bin/bash
# Builds the standalone GPU probes under tools/bin/ (git-ignored; they travel to the GPU box with gpurun):
#   bw_probe       streaming / boundary / latency probe
#   decode_bench   decode-step microbenchmark against the in-tree library
#   trace/         the library rebuilt with -DDOTS_TRACE + decode_bench_trace (per-phase timestamps inside the decode kernels)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/trace
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Idots_ocr_amd/csrc -Iinclude"
python -c "from dots_ocr_amd import build; build.build()"
hipcc $FLAGS tools/bw_probe.hip -o tools/bin/bw_probe 2>/dev/null
hipcc $FLAGS tools/decode_bench.hip -Ldots_ocr_amd/lib -ldots_ocr_hip -Wl,-rpath,'$ORIGIN/../../dots_ocr_amd/lib' -o tools/bin/decode_bench 2>/dev/null
if [ "$1" = "trace" ]; then
  for f in dots_ocr_amd/csrc/*.hip; do hipcc $FLAGS -fPIC -DDOTS_TRACE -x hip -c $f -o tools/bin/trace/$(basename $f).o 2>/dev/null & done; wait
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib tools/bin/trace/*.o -o tools/bin/trace/libdots_ocr_hip.so
  hipcc $FLAGS -DDOTS_TRACE tools/decode_bench.hip -Ltools/bin/trace -ldots_ocr_hip -Wl,-rpath,'$ORIGIN/trace' -o tools/bin/decode_bench_trace 2>/dev/null
fi
ls -la tools/bin
