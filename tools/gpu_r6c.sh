# Round 6, GPU call C: the driver's own bench command (page-set rotation, h2d, other_configs legs) with its wall time, then the whole GPU suite with per-test durations.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6c; mkdir -p $O
T0=$(date +%s)
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
python - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("a4 value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items()}, d.get("parity_vs_sequential"), "sets", d.get("page_sets"), "h2d", d.get("h2d",{}).get("ms_per_step"), d.get("h2d",{}).get("pcie_inclusive_pages_per_s"))
print("decode frac", d["roofline_decode"]["frac"], "seq decode frac", d["roofline_decode_sequential"]["frac"], "vit seq", d["roofline_vit_sequential"]["frac"], "attn seq", d["roofline_sequential"]["frac"], "tail", d["overlap"]["tower_tail_blocks"])
for k,v in d.get("other_configs",{}).items():
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","failed","rc","leg_wall_s","parity_vs_sequential","parity_vs_single_sequence","stderr_tail","error")})
print("cpu_baseline", d.get("cpu_baseline",{}).get("value"))
PY
tail -3 $O/bench_default.err
T1=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=60 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$? wall=$(( $(date +%s) - T1 )) s"
tail -75 $O/pytest_gpu.log | cut -c1-200
