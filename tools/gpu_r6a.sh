# Round 6, GPU call A: the four-tile gate|up / lm_head kernels (csrc/decode_b64.hip): bitwise parity vs the one-tile kernels for every launch
# shape, then decode_bench at 64 rows on the 64-CU partition / whole chip, new kernels vs the round-4 two-tile kernels (DOTS_OCR_DEC_S64=0).
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6a; mkdir -p $O
timeout 1200 python -m pytest tests/test_decode_kernels_gpu.py -x -q -m gpu -k "four_tile or two_tile" > $O/pytest_tiles.log 2>&1; echo "four-tile + two-tile pytest rc=$?"; tail -5 $O/pytest_tiles.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan, four-tile gate|up + lm_head" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "64 rows, 64-CU partition plan, round-4 two-tile kernels" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_S64=0
db "64 rows, whole chip, four-tile gate|up + lm_head" X=1
db "64 rows, whole chip, round-4 two-tile kernels" DOTS_OCR_DEC_S64=0
db "64 rows, 96-CU partition plan, four-tile" DOTS_BENCH_CUS=96 DOTS_BENCH_FULL=1
db "64 rows, 128-CU partition plan, four-tile" DOTS_BENCH_CUS=128 DOTS_BENCH_FULL=1
db "64 rows, 128-CU partition plan, two-tile" DOTS_BENCH_CUS=128 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_S64=0
ARGS="48 5700 6288"
db "48 rows, 64-CU partition plan, four-tile" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "48 rows, 64-CU partition plan, two-tile" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_S64=0
grep -E "^==|whole step|dec_gateup  |dec_lmhead  |marginal dec_gateup|marginal dec_lmhead" $O/decode_bench.txt
