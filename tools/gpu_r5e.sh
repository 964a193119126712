# Round 5, GPU call E: fast silu + plan 1 as candidate default: parity subset, gemm A/B, end-to-end bench under both plans
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5e; mkdir -p $O
timeout 500 python -m pytest tests/test_gemm_plans_gpu.py -x -q -m gpu > $O/pytest_plans.log 2>&1; echo "plans pytest rc=$?"; tail -3 $O/pytest_plans.log
DOTS_OCR_GEMM_PLAN=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_vit_parity_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu > $O/pytest_plan1.log 2>&1; echo "kernels + fullsize parity (plan 1) rc=$?"; tail -3 $O/pytest_plan1.log
run() { ( echo "== $1"; shift; env "$@" timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_silu.txt; }
run "plan 1, fast silu" DOTS_OCR_GEMM_PLAN=1
run "plan 1, exact silu (-DGEMM_EXACT_SILU)" DOTS_OCR_GEMM_PLAN=1 DOTS_OCR_LIB=$R/tools/bin/var_w4_exact_silu/libdots_ocr_hip.so
cat $O/gemm_silu.txt
for p in 0 1; do
  ( DOTS_OCR_GEMM_PLAN=$p timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/bench_plan$p.log 2>&1; grep '^{"metric"' $O/bench_plan$p.log | tail -1 > $O/bench_plan$p.json
  python - $O/bench_plan$p.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "roofline", d.get("roofline",{}).get("frac"), d.get("parity_vs_sequential"))
except Exception as e: print("FAILED", e)
PY
done
tail -5 $O/bench_plan1.log | cut -c1-400
