# Round 5, GPU call F (re-entry session): the tests that have never run on a GPU (planted walk, flash sampled rows above A4, fp8 two-tile,
# streaming decode attention), then the streaming decode-attention kernel against the per-split kernel: decode_bench at 64 rows on the
# 64-CU partition / on the whole chip / at 8 rows, two variant builds, and the a4 bench line with and without it.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_decode_kernels_gpu.py -x -q -m gpu -k "attention or two_tile" > $O/pytest_attn.log 2>&1; echo "attention + two-tile pytest rc=$?"; tail -3 $O/pytest_attn.log
timeout 600 python -m pytest tests/test_decode_plans_gpu.py -x -q -m gpu > $O/pytest_plans.log 2>&1; echo "decode plans pytest rc=$?"; tail -3 $O/pytest_plans.log
timeout 900 python -m pytest tests/test_planted_walk_gpu.py "tests/test_fullsize_gpu.py::test_flash_attn64_sampled_rows_match_oracle_above_a4" "tests/test_fullsize_gpu.py::test_flash_attn64_ragged_packed_batch_of_the_mixed64_sizes" -x -q -m gpu > $O/pytest_new.log 2>&1; echo "planted walk + flash above A4 pytest rc=$?"; tail -3 $O/pytest_new.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan, per-split attention" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_ATTN_STREAM=0
db "64 rows, 64-CU partition plan, streaming attention" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_ATTN_STREAM=1
db "64 rows, 64-CU partition plan, streaming attention, plain (allocating) DMA" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_ATTN_STREAM=1 LD_LIBRARY_PATH=$R/tools/bin/var_attn_aux0
db "64 rows, 64-CU partition plan, streaming attention, K half requested early" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_ATTN_STREAM=1 LD_LIBRARY_PATH=$R/tools/bin/var_attn_split
db "64 rows, whole chip, per-split attention" DOTS_OCR_ATTN_STREAM=0
db "64 rows, whole chip, streaming attention" DOTS_OCR_ATTN_STREAM=1
ARGS="8 5700 6288"
db "8 rows, whole chip, per-split attention" DOTS_OCR_ATTN_STREAM=0
db "8 rows, whole chip, streaming attention" DOTS_OCR_ATTN_STREAM=1
ARGS="16 5700 6288"
db "16 rows, 64-CU partition plan, per-split attention" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_ATTN_STREAM=0
db "16 rows, 64-CU partition plan, streaming attention" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_ATTN_STREAM=1
grep -E "^==|whole step|^decode_attn |decode attention|dec_proj down|dec_gateup  |dec_qkv  " $O/decode_bench.txt
for m in 0 auto; do
  if [ $m = auto ]; then E=""; else E="DOTS_OCR_ATTN_STREAM=$m"; fi
  ( env $E timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > $O/bench_stream_$m.log 2>&1; grep '^{"metric"' $O/bench_stream_$m.log | tail -1 > $O/bench_stream_$m.json
  python - $O/bench_stream_$m.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"))
except Exception as e: print("FAILED", e)
PY
done
tail -3 $O/bench_stream_auto.log | cut -c1-300
