# Round 6, GPU call N: down_proj above 32 rows as two K halves (decode_b64.hip): bitwise tests, decode_bench A/B (DOTS_OCR_DEC_KHALF=0 = the full-K wide kernel), bench parity
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6n; mkdir -p $O
timeout 1500 python -m pytest tests/test_decode_kernels_gpu.py tests/test_decode_plans_gpu.py tests/test_fp8_gpu.py -x -q -m gpu > $O/pytest_decode.log 2>&1; echo "decode kernels + plans + fp8 pytest rc=$?"; tail -3 $O/pytest_decode.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan, K-half down_proj" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "64 rows, 64-CU partition plan, full-K wide down_proj" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_KHALF=0
db "64 rows, whole chip, K-half down_proj" X=1
db "64 rows, whole chip, full-K wide down_proj" DOTS_OCR_DEC_KHALF=0
db "64 rows, 128-CU partition plan, K-half down_proj" DOTS_BENCH_CUS=128 DOTS_BENCH_FULL=1
db "64 rows, 128-CU partition plan, full-K" DOTS_BENCH_CUS=128 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_KHALF=0
grep -E "^==|whole step|dec_proj down  |dec_qkv  |marginal dec_proj down|marginal dec_qkv|dec_lmhead  " $O/decode_bench.txt
( DOTS_BENCH_OTHER=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/bench_a4.log 2>&1; grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/bench_a4.json
( timeout 600 python bench.py --workload highres --batch 4 --steps 8 --warmup 4 --no-cpu-baseline ) > $O/bench_highres_b4.log 2>&1; grep '^{"metric"' $O/bench_highres_b4.log | tail -1 > $O/bench_highres_b4.json
python - $O/bench_a4.json $O/bench_highres_b4.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items() if "ms" in k and "ize" not in k and "process" not in k}, "decode frac", round(d["roofline_decode"]["frac"],4), d.get("parity_vs_sequential"))
    except Exception as e: print(f, "FAILED", e)
PY
tail -2 $O/bench_a4.log | cut -c1-300
