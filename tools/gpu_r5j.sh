# Round 5, GPU call J: wide kernels second pass (one-round o_proj, earlier X loads in qkv): bitwise tests, decode_bench at 64 rows; mixed64 with the
# horizon-aware look-ahead; a4 line.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5j; mkdir -p $O
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "tail", d.get("overlap",{}).get("tower_tail_blocks"), "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"), d.get("parity_vs_single_sequence"))
    except Exception as e: print(f, "FAILED", e)
PY
}
timeout 600 python -m pytest tests/test_decode_kernels_gpu.py tests/test_decode_plans_gpu.py -x -q -m gpu -k "wide or plans or plan or dec_qkv or dec_proj" > $O/pytest_wide.log 2>&1; echo "wide + plans pytest rc=$?"; tail -2 $O/pytest_wide.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan, wide kernels (second pass)" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "64 rows, whole chip, wide kernels (second pass)"
ARGS="32 5700 6288"
db "32 rows, 96-CU partition plan, wide kernels (second pass)" DOTS_BENCH_CUS=96 DOTS_BENCH_FULL=1
grep -E "^==|whole step|^decode_attn |dec_proj o  |dec_proj down|dec_gateup  |dec_qkv  " $O/decode_bench.txt
( timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_mixed64.log 2>&1; grep '^{"metric"' $O/bench_mixed64.log | tail -1 > $O/bench_mixed64.json
( timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/bench_a4.log 2>&1; grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/bench_a4.json
line $O/bench_mixed64.json $O/bench_a4.json
tail -2 $O/bench_mixed64.log | cut -c1-200
