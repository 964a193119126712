# Round 5, closing session: the whole -m gpu suite on the FINAL tree (what the driver runs at round end) with durations, smoke(), and the
# default bench command once more (the driver's line).
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5m; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -18 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( timeout 700 python bench.py --steps 20 --warmup 2 ) > $O/bench_a4.log 2>&1; grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/r05_bench_a4_20steps.json
python - $O/r05_bench_a4_20steps.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("a4 20 steps: value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "tail", d.get("overlap",{}).get("tower_tail_blocks"), d.get("parity_vs_sequential"), d.get("steps_checked"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
