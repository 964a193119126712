# Round 6, GPU call Y: bench.py with 4 vs 8 waves per decode-attention workgroup (a4 line: roofline_decode_sequential at B = 8, the 64-row partition step, svg leg at B = 1)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6y; mkdir -p $O
for wv in 4 8; do
( DOTS_OCR_ATTN_WAVES=$wv timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs ) > $O/bench_w$wv.log 2>&1; grep '^{"metric"' $O/bench_w$wv.log | tail -1 > $O/bench_w$wv.json
( DOTS_OCR_ATTN_WAVES=$wv timeout 900 python bench.py --workload svg --fp8 1 --steps 1 --warmup 1 --no-cpu-baseline ) > $O/svg_w$wv.log 2>&1; grep '^{"metric"' $O/svg_w$wv.log | tail -1 > $O/svg_w$wv.json
python - $O/bench_w$wv.json $O/svg_w$wv.json $wv <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("waves", sys.argv[3], "a4 %.3f"%d["value"], "decode_ms %.1f"%d["phase_ms_per_step"]["decode_ms"], "seq decode frac %.4f"%d["roofline_decode_sequential"]["frac"], "alone 64 rows %.4f"%d["roofline_decode_alone_rows_in_flight"]["frac"], "seq pages/s %.3f"%d["throughput_shapes"]["sequential_batch"]["pages_per_s"], d.get("parity_vs_sequential"))
try:
    s=json.load(open(sys.argv[2])); print("   svg %.4f pages/s ms/step %.1f"%(s["value"], s["ms_per_step"]), "decode frac %.4f"%s["roofline"]["frac"])
except Exception as e: print("   svg failed", e)
PY
done
