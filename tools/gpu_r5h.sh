# Round 5, GPU call H: the adaptive tower tail (A/B against tail 0 and a fixed tail), the B = 8 in-kernel timeline of the decode step
# (VERDICT r4 #2), highres + mixed64 on the wide-kernel build.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5h; mkdir -p $O
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "tail", d.get("overlap",{}).get("tower_tail_blocks"), "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"), d.get("parity_vs_single_sequence"))
    except Exception as e: print(f, "FAILED", e)
PY
}
for t in auto 0 7; do
  if [ $t = auto ]; then E=""; else E="DOTS_OCR_TOWER_TAIL_LAYERS=$t"; fi
  ( env $E DOTS_BENCH_TRACE=1 timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > $O/bench_tail_$t.log 2>&1; grep '^{"metric"' $O/bench_tail_$t.log | tail -1 > $O/bench_tail_$t.json
  line $O/bench_tail_$t.json
done
grep "^\[step" $O/bench_tail_auto.log | tail -4 | cut -c1-260
( timeout 120 tools/bin/decode_bench_trace 8 5200 6288 2>&1 | grep -v amdgpu.ids ) > $O/decode_trace_b8.txt; tail -40 $O/decode_trace_b8.txt
( timeout 300 python bench.py --workload highres --batch 4 --no-cpu-baseline ) > $O/bench_highres.log 2>&1; grep '^{"metric"' $O/bench_highres.log | tail -1 > $O/bench_highres.json
( timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_mixed64.log 2>&1; grep '^{"metric"' $O/bench_mixed64.log | tail -1 > $O/bench_mixed64.json
line $O/bench_highres.json $O/bench_mixed64.json
tail -2 $O/bench_mixed64.log | cut -c1-400
