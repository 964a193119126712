# Round 6, GPU call D: the LM prefill on the 64-row flash kernel (causal instantiation): parity tests, then bench A/B of the prefill phase.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "kernels + fullsize + model pytest rc=$?"; tail -4 $O/pytest.log
for m in 1 0; do
  ( DOTS_OCR_PREFILL_F64=$m DOTS_BENCH_OTHER=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/bench_f64_$m.log 2>&1; grep '^{"metric"' $O/bench_f64_$m.log | tail -1 > $O/bench_f64_$m.json
  python - $O/bench_f64_$m.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items()}, d.get("parity_vs_sequential"), "seq step", d["overlap"]["sequential_step_ms_same_run"])
except Exception as e: print("FAILED", e)
PY
done
