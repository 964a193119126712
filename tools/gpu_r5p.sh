# Round 5, GPU call P: flash attention work list cut by COST across the XCDs (XcdPlan): flash tests (uniform, above A4, ragged packed batch, tiny
# shapes), the mixed64 line (ragged towers: the imbalance was 1.32 x) and the a4 line (uniform: must be unchanged).
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5p; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "flash or attn or attention" > $O/pytest_flash.log 2>&1; echo "flash pytest rc=$?"; tail -2 $O/pytest_flash.log
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "roofline", d.get("roofline",{}).get("frac"), d.get("parity_vs_sequential"), d.get("parity_vs_single_sequence"))
    except Exception as e: print(f, "FAILED", e)
PY
}
( timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r05_bench_mixed64.json; line $O/r05_bench_mixed64.json
( timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_a4.json; line $O/bench_a4.json
