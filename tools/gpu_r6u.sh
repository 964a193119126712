# Round 6, GPU call U: the bench's decode-alone measurement (the timed region's decode step, every slot group full, alone on the chip)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6u; mkdir -p $O
( DOTS_BENCH_OTHER=0 timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs ) > $O/bench_short.log 2>&1; echo "bench rc=$?"; tail -3 $O/bench_short.log | cut -c1-400
grep '^{"metric"' $O/bench_short.log | tail -1 > $O/bench_short.json
python - $O/bench_short.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("a4", round(d["value"],3), d.get("parity_vs_sequential"))
print(json.dumps(d.get("roofline_decode_alone_rows_in_flight"), indent=1)); print(d["targets"])
PY
