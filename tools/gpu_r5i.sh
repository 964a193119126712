# Round 5, GPU call I: mixed64 regression hunt (wide kernels on / off), the tower started beside the prefill (TOWER_NOW) with the adaptive
# tail, the in-kernel timeline of the wide kernels at 64 rows on the 64-CU partition, rope-split test.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5i; mkdir -p $O
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "tail", d.get("overlap",{}).get("tower_tail_blocks"), "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"), d.get("parity_vs_single_sequence"))
    except Exception as e: print(f, "FAILED", e)
PY
}
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rope or split" > $O/pytest_rope.log 2>&1; echo "rope pytest rc=$?"; tail -2 $O/pytest_rope.log
( timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_mixed64.log 2>&1; grep '^{"metric"' $O/bench_mixed64.log | tail -1 > $O/bench_mixed64.json
( DOTS_OCR_DEC_WIDE=0 timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_mixed64_tile.log 2>&1; grep '^{"metric"' $O/bench_mixed64_tile.log | tail -1 > $O/bench_mixed64_tile.json
( DOTS_OCR_GEMM_PLAN=0 timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_mixed64_gemm0.log 2>&1; grep '^{"metric"' $O/bench_mixed64_gemm0.log | tail -1 > $O/bench_mixed64_gemm0.json
line $O/bench_mixed64.json $O/bench_mixed64_tile.json $O/bench_mixed64_gemm0.json
( DOTS_BENCH_TOWER_NOW=1 DOTS_BENCH_TRACE=1 timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/bench_tower_now.log 2>&1; grep '^{"metric"' $O/bench_tower_now.log | tail -1 > $O/bench_tower_now.json
( DOTS_BENCH_TRACE=1 timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
line $O/bench_tower_now.json $O/bench_default.json
grep "^\[step" $O/bench_tower_now.log | tail -2 | cut -c1-260
( DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 120 tools/bin/decode_bench_trace 64 5700 6288 2>&1 | grep -v amdgpu.ids ) > $O/decode_trace_b64_part.txt; grep -A1 "^dec_qkv\|^dec_proj" $O/decode_trace_b64_part.txt
