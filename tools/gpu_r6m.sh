# Round 6, GPU call M: highres with 64 / 96 / 128 decode CUs (the four-tile kernels made the 64-row step cheaper on small partitions), a4 with --rows-in-flight 8 (throughput_shapes)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6m; mkdir -p $O
run() { tag=$1; shift; ( env "$@" ) > $O/$tag.log 2>&1; grep '^{"metric"' $O/$tag.log | tail -1 > $O/$tag.json
  python - $O/$tag.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items() if k in ("vit_ms","prefill_ms","decode_ms","vit_attn_ms")}, "tail", d.get("overlap",{}).get("tower_tail_blocks"), d.get("parity_vs_sequential"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
H="timeout 400 python bench.py --workload highres --steps 8 --warmup 4 --no-cpu-baseline"
run highres_c128 $H
run highres_c96 DOTS_OCR_OVERLAP_DEC_CUS=96 $H
run highres_c64 DOTS_OCR_OVERLAP_DEC_CUS=64 $H
run highres_c128_b8 $H --batch 8
run highres_c64_b8 DOTS_OCR_OVERLAP_DEC_CUS=64 $H --batch 8
run a4_one_batch_in_flight DOTS_BENCH_OTHER=0 timeout 400 python bench.py --rows-in-flight 8 --no-cpu-baseline
cp $O/a4_one_batch_in_flight.json $O/r06_bench_a4_one_batch_in_flight.json
