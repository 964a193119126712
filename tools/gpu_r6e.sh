# Round 6, GPU call E: dec_proj_wide as one rolling stream per K slice: bitwise test, decode_bench at 64 / 48 / 32 rows.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6e; mkdir -p $O
timeout 1500 python -m pytest tests/test_decode_kernels_gpu.py tests/test_decode_plans_gpu.py -x -q -m gpu -k "wide_kernels or plans" > $O/pytest_wide.log 2>&1; echo "wide + plans pytest rc=$?"; tail -3 $O/pytest_wide.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "64 rows, whole chip" X=1
db "64 rows, 128-CU partition plan" DOTS_BENCH_CUS=128 DOTS_BENCH_FULL=1
ARGS="32 5700 6288"
db "32 rows, 64-CU partition plan" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "32 rows, whole chip" X=1
ARGS="8 5700 6288"
db "8 rows, whole chip" X=1
grep -E "^==|whole step|dec_proj down  |dec_proj o   |marginal dec_proj" $O/decode_bench.txt
