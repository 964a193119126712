# two-tile gate|up / lm_head (batches above 16 rows): parity tests, then the bench lines they move
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4q; mkdir -p $O
timeout 500 python -m pytest tests/test_decode_kernels_gpu.py tests/test_decode_plans_gpu.py tests/test_overlap_gpu.py tests/test_fp8_gpu.py -x -q -m gpu > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $O/pytest_kernels.log
timeout 500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "batch or continuous or server or slots" > $O/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -3 $O/pytest_model.log
run() { name=$1; shift; ( timeout 400 "$@" ) > $O/$name.log 2>&1; grep '^{"metric"' $O/$name.log | tail -1 > $O/$name.json; }
run a4 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
DOTS_OCR_GATEUP_PER_TILE=1 DOTS_OCR_LMHEAD_PER_TILE=1 run a4_per_tile python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run highres python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline
run mixed64 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, d.get("parity_vs_sequential"), d.get("steps_checked"))
    except Exception as e: print(f, "FAILED", e); print(open(f.replace(".json",".log")).read()[-1500:])
PY
