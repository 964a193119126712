# Round 5, GPU call G: the WIDE decode kernels (qkv + projections above 16 rows): bitwise tests against the per-tile kernels, decode plans,
# the fixed packed-batch flash test; decode_bench at 64 rows (64-CU partition plan / whole chip) wide vs per-tile; the a4 bench line with
# the decode partition at 64 and 56 CUs.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py -x -q -m gpu -k "wide or dec_qkv or dec_proj or reject" > $O/pytest_wide.log 2>&1; echo "wide kernels pytest rc=$?"; tail -3 $O/pytest_wide.log
timeout 600 python -m pytest tests/test_decode_plans_gpu.py "tests/test_fullsize_gpu.py::test_flash_attn64_ragged_packed_batch_of_the_mixed64_sizes" -x -q -m gpu > $O/pytest_plans.log 2>&1; echo "decode plans + packed flash pytest rc=$?"; tail -3 $O/pytest_plans.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan, wide kernels" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "64 rows, 64-CU partition plan, per-tile kernels" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_WIDE=0
db "64 rows, 56-CU partition plan, wide kernels" DOTS_BENCH_CUS=56 DOTS_BENCH_FULL=1
db "64 rows, whole chip, wide kernels"
db "64 rows, whole chip, per-tile kernels" DOTS_OCR_DEC_WIDE=0
ARGS="32 5700 6288"
db "32 rows, 64-CU partition plan, wide kernels" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "32 rows, 64-CU partition plan, per-tile kernels" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_WIDE=0
grep -E "^==|whole step|^decode_attn |dec_proj o  |dec_proj down|dec_gateup  |dec_qkv  " $O/decode_bench.txt
for cus in 64 56; do
  ( DOTS_OCR_OVERLAP_DEC_CUS=$cus timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > $O/bench_cus$cus.log 2>&1; grep '^{"metric"' $O/bench_cus$cus.log | tail -1 > $O/bench_cus$cus.json
  python - $O/bench_cus$cus.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"))
except Exception as e: print("FAILED", e)
PY
done
tail -3 $O/bench_cus56.log | cut -c1-300
