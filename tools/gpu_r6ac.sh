# Round 6, GPU call AC: the other bench shapes on the last commit (one batch in flight, strictly sequential, 16 rows in flight, page queue)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6ac; mkdir -p $O
run() { n=$1; shift; ( DOTS_BENCH_OTHER=0 timeout 600 python bench.py --no-cpu-baseline "$@" ) > $O/$n.log 2>&1; rc=$?; grep '^{"metric"' $O/$n.log | tail -1 > $O/$n.json
python - $O/$n.json $n $rc <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "rc", sys.argv[3], "value %.3f"%d["value"], "ms/step %.1f"%d["ms_per_step"], d.get("parity_vs_sequential") or d.get("parity_vs_single_sequence"), "alone" , (d.get("roofline_decode_alone_rows_in_flight") or {}).get("frac"))
except Exception as e: print(sys.argv[2], "rc", sys.argv[3], "FAILED", e)
PY
}
run one_batch --rows-in-flight 8 --steps 3 --warmup 1
run sequential --no-overlap --steps 2 --warmup 1
run rows16 --rows-in-flight 16 --steps 3 --warmup 2
run mixed_queue --workload mixed64 --page-queue --steps 1 --warmup 0
tail -3 $O/*.log | grep -i "error\|Traceback" | head
