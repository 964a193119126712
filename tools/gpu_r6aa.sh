# Round 6, GPU call AA: a4 decode-partition sweep on the final build (the decode step got 13 % faster this round: did the optimum move?)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6aa; mkdir -p $O; rm -f $O/sweep.txt
for cus in 64 48 56 72 80 96 64; do
  ( DOTS_OCR_OVERLAP_DEC_CUS=$cus DOTS_BENCH_DECODE_ALONE=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs ) > $O/b_$cus.log 2>&1
  grep '^{"metric"' $O/b_$cus.log | tail -1 > $O/b_$cus.json
  python - $O/b_$cus.json $cus <<'PY' >> $O/sweep.txt
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("decode CUs %3s: %.3f pages/s  step %.1f ms"%(sys.argv[2], d["value"], d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items() if k in ("vit_ms","prefill_ms","decode_ms")}, "tail", d["overlap"]["tower_tail_blocks"], d.get("parity_vs_sequential"))
except Exception as e: print("decode CUs", sys.argv[2], "FAILED", e)
PY
done
cat $O/sweep.txt
