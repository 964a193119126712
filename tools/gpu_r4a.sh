# GPU run r4a: decode kernels of round 4 (attention register diet + conflict-free tail, pair-walking gate|up) — parity subset, microbench, traces, one short bench
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4a; mkdir -p $O
( timeout 900 python -m pytest tests/test_decode_kernels_gpu.py tests/test_flow_gpu.py -x -q 2>&1 | tail -15 ) > $O/pytest_decode.log 2>&1
( timeout 120 tools/bin/decode_bench 8 5700 6288 ) > $O/decode_bench_whole.txt 2>&1
( DOTS_BENCH_FULL=1 DOTS_BENCH_CUS=128 timeout 120 tools/bin/decode_bench 8 5700 6288 ) > $O/decode_bench_128_full.txt 2>&1
( DOTS_BENCH_FULL=1 DOTS_BENCH_CUS=96 timeout 120 tools/bin/decode_bench 8 5700 6288 ) > $O/decode_bench_96_full.txt 2>&1
( DOTS_BENCH_FULL=1 timeout 120 tools/bin/decode_bench 8 5700 6288 ) > $O/decode_bench_whole_full.txt 2>&1
( timeout 120 tools/bin/decode_bench 1 500 4700 ) > $O/decode_bench_b1.txt 2>&1
( DOTS_BENCH_FP8=1 timeout 120 tools/bin/decode_bench 1 500 4700 ) > $O/decode_bench_b1_fp8.txt 2>&1
( timeout 120 tools/bin/decode_bench 16 5700 6288 ) > $O/decode_bench_b16.txt 2>&1
( DOTS_BENCH_FULL=1 DOTS_BENCH_CUS=128 timeout 120 tools/bin/decode_bench 16 5700 6288 ) > $O/decode_bench_b16_128.txt 2>&1
( timeout 120 tools/bin/decode_bench_trace 8 5700 6288 ) > $O/decode_trace_whole.txt 2>&1
( DOTS_BENCH_FULL=1 DOTS_BENCH_CUS=128 timeout 120 tools/bin/decode_bench_trace 8 5700 6288 ) > $O/decode_trace_128.txt 2>&1
( timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_a4.log 2>&1
grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/bench_a4.json
tail -3 $O/pytest_decode.log; grep "whole step" $O/decode_bench_*.txt; python - <<'PY'
import json,os
p=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4a/bench_a4.json'
try:
    d=json.load(open(p)); print({k:d.get(k) for k in ['value','ms_per_step','parity_vs_sequential','steps_checked']}, d.get('targets'))
except Exception as e: print('bench json missing', e)
PY
