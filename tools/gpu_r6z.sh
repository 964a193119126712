# Round 6, GPU call Z: the decode-alone measurement on the other shapes (highres batch 4, short generations), then the driver's bench command again
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6z; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get("roofline_decode_alone_rows_in_flight")
    print(sys.argv[1].split("/")[-1], "value %.3f"%d["value"], d.get("parity_vs_sequential"), "alone:", None if r is None else {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ("rows","steps","ms_per_step","frac","mean_context")})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
( timeout 600 python bench.py --workload highres --batch 4 --steps 8 --warmup 4 --no-cpu-baseline ) > $O/hr.log 2>&1; echo "highres rc=$?"; grep "skipped" $O/hr.log; grep '^{"metric"' $O/hr.log | tail -1 > $O/hr.json; show $O/hr.json
( DOTS_BENCH_OTHER=0 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 128 ) > $O/short128.log 2>&1; echo "a4 128 tokens rc=$?"; grep "skipped" $O/short128.log; grep '^{"metric"' $O/short128.log | tail -1 > $O/short128.json; show $O/short128.json
( DOTS_BENCH_OTHER=0 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 16 ) > $O/short16.log 2>&1; echo "a4 16 tokens rc=$?"; grep "skipped" $O/short16.log; grep '^{"metric"' $O/short16.log | tail -1 > $O/short16.json; show $O/short16.json
T0=$(date +%s)
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
python - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("a4 value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items()}, d.get("parity_vs_sequential"), "sets", d.get("page_sets"))
print("roofline", round(d["roofline"]["frac"],4), "vit", round(d["roofline_vit"]["frac"],4), "decode", round(d["roofline_decode"]["frac"],4), "| seq: attn", round(d["roofline_sequential"]["frac"],4), "vit", round(d["roofline_vit_sequential"]["frac"],4), "decode", round(d["roofline_decode_sequential"]["frac"],4), "alone rows", round(d["roofline_decode_alone_rows_in_flight"]["frac"],4), "tail", d["overlap"]["tower_tail_blocks"])
for k,v in d.get("other_configs",{}).items():
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","failed","rc","leg_wall_s","parity_vs_sequential","parity_vs_single_sequence","stderr_tail","error")})
PY
