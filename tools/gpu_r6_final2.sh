# Round 6, last validation call: rocprofv3 kernel stats (sequential and pipelined a4 command) on the final build, then the driver's bench command
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6p; mkdir -p $O/prof
for mode in seq ovl; do
  rm -rf $O/prof/*
  if [ $mode = seq ]; then FL="--no-overlap --max-new-tokens 64"; else FL="--max-new-tokens 128"; fi
  DOTS_BENCH_OTHER=0 timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o r06 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $FL > $O/prof/bench.log 2>&1; echo "rocprof $mode rc=$?"
  grep '^{"metric"' $O/prof/bench.log | tail -1 > $O/r06_a4_b8_${mode}_bench_line.json
  db=$(find $O/prof -name "*.db" | head -1)
  python profiles/summarize_rocprof.py $db > $O/r06_a4_b8_${mode}_kernel_stats.txt; head -6 $O/r06_a4_b8_${mode}_kernel_stats.txt | cut -c1-150
done
rm -rf $O/prof
O=$R/gpurun_out/r6z; mkdir -p $O
T0=$(date +%s)
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
python - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("a4 value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items()}, d.get("parity_vs_sequential"), "sets", d.get("page_sets"), "h2d", d.get("h2d",{}).get("ms_per_step"))
print("roofline", round(d["roofline"]["frac"],4), "vit", round(d["roofline_vit"]["frac"],4), "decode", round(d["roofline_decode"]["frac"],4), "| seq: attn", round(d["roofline_sequential"]["frac"],4), "vit", round(d["roofline_vit_sequential"]["frac"],4), "decode", round(d["roofline_decode_sequential"]["frac"],4), "alone rows", round(d["roofline_decode_alone_rows_in_flight"]["frac"],4), "tail", d["overlap"]["tower_tail_blocks"])
for k,v in d.get("other_configs",{}).items():
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","failed","rc","leg_wall_s","parity_vs_sequential","parity_vs_single_sequence","stderr_tail","error")})
print("cpu_baseline", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"))
PY
