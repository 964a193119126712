# Round 6, GPU call B: qkv above 32 rows on the X image of the norm kernel (bitwise test), decode_bench A/B, then the bench lines (a4, highres).
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_decode_kernels_gpu.py -x -q -m gpu -k "wide_kernels" > $O/pytest_wide.log 2>&1; echo "wide pytest rc=$?"; tail -3 $O/pytest_wide.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
db "64 rows, 64-CU partition plan, qkv on the X image" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1
db "64 rows, 64-CU partition plan, qkv normalising in the kernel" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_QKV_XIMG=0
db "64 rows, whole chip, qkv on the X image" X=1
db "64 rows, whole chip, qkv normalising in the kernel" DOTS_OCR_QKV_XIMG=0
grep -E "^==|whole step|dec_qkv  |marginal dec_qkv" $O/decode_bench.txt
for w in a4 highres; do
  ( timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline ) > $O/bench_$w.log 2>&1; grep '^{"metric"' $O/bench_$w.log | tail -1 > $O/bench_$w.json
  python - $O/bench_$w.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"))
except Exception as e: print("FAILED", e)
PY
  tail -2 $O/bench_$w.log | cut -c1-400
done
