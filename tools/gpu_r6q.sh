# Round 6, GPU call Q: after the counted-wait fix + prologue reorder of dec_stream64: four-tile bitwise tests, model tests (small model: L = 6), decode_bench 64 rows, a4 / highres bench
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6q; mkdir -p $O
timeout 1500 python -m pytest tests/test_decode_kernels_gpu.py tests/test_model_gpu.py tests/test_overlap_gpu.py tests/test_a4_anchor_gpu.py -x -q -m gpu -k "four_tile or two_tile or model or overlap or anchor or batch or slots or generate" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "64 rows, whole chip" X=1
db "64 rows, 128-CU partition plan" DOTS_BENCH_CUS=128 DOTS_BENCH_FULL=1
ARGS="48 5700 6288"
db "48 rows, whole chip" X=1
grep -E "^==|whole step|dec_proj down  |dec_proj o   |dec_qkv  |dec_gateup  |dec_lmhead  |decode_attn " $O/decode_bench.txt
( DOTS_BENCH_OTHER=0 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_a4.log 2>&1; grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/bench_a4.json
( timeout 600 python bench.py --workload highres --batch 4 --steps 8 --warmup 4 --no-cpu-baseline ) > $O/bench_highres_b4.log 2>&1; grep '^{"metric"' $O/bench_highres_b4.log | tail -1 > $O/bench_highres_b4.json
python - $O/bench_a4.json $O/bench_highres_b4.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items() if "ms" in k and "ize" not in k and "process" not in k}, "decode frac", round(d["roofline_decode"]["frac"],4), d.get("parity_vs_sequential"))
    except Exception as e: print(f, "FAILED", e)
PY
