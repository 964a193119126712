# Round 6, final validation: the driver's GPU test command with durations, smoke(), the driver's bench command, fp8 a4 sanity, then the profiling pass on the final build
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6z; mkdir -p $O
T1=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=30 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$? wall=$(( $(date +%s) - T1 )) s"
grep -E "passed|failed|skipped" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-300
T0=$(date +%s)
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
python - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("a4 value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items()}, d.get("parity_vs_sequential"), "sets", d.get("page_sets"), "h2d", d.get("h2d",{}).get("ms_per_step"))
print("roofline", round(d["roofline"]["frac"],4), "vit", round(d["roofline_vit"]["frac"],4), "decode", round(d["roofline_decode"]["frac"],4), "| seq: attn", round(d["roofline_sequential"]["frac"],4), "vit", round(d["roofline_vit_sequential"]["frac"],4), "decode", round(d["roofline_decode_sequential"]["frac"],4), "tail", d["overlap"]["tower_tail_blocks"], "traffic", d["roofline"]["traffic"], d["roofline_decode"]["traffic"])
for k,v in d.get("other_configs",{}).items():
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","failed","rc","leg_wall_s","parity_vs_sequential","parity_vs_single_sequence","stderr_tail","error")})
print("cpu_baseline", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"))
PY
( DOTS_BENCH_OTHER=0 timeout 600 python bench.py --fp8 1 --steps 4 --warmup 2 --no-cpu-baseline ) > $O/bench_a4_fp8.log 2>&1; grep '^{"metric"' $O/bench_a4_fp8.log | tail -1 > $O/bench_a4_fp8.json
python - $O/bench_a4_fp8.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("a4 fp8 weights: value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items() if "ize" not in k and "process" not in k}, d.get("parity_vs_sequential"))
except Exception as e: print("fp8 a4 FAILED", e)
PY
bash tools/gpu_r6_profiles.sh > $O/profiles.log 2>&1; grep -E "rc=|total kernel time" $O/profiles.log
