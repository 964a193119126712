# Round 6, GPU call G: mixed64 with 32 / 48 / 64 slots and 64 / 96 / 128 decode CUs (the four-tile kernels make rows above 32 cheap), --page-queue at N = 1
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6g; mkdir -p $O
run() { tag=$1; shift; ( env "$@" ) > $O/$tag.log 2>&1; grep '^{"metric"' $O/$tag.log | tail -1 > $O/$tag.json
  python - $O/$tag.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.4f ms/step %.1f"%(d["value"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items() if k in ("vit_ms","decode_ms")}, "dec frac", round(d.get("roofline_decode",{}).get("frac",0),3), d.get("parity_vs_single_sequence"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
M="timeout 400 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline"
run mixed_b32_c64 $M
run mixed_b64_c64 $M --batch 64
run mixed_b48_c64 $M --batch 48
run mixed_b64_c96 DOTS_OCR_OVERLAP_DEC_CUS=96 $M --batch 64
run mixed_b64_c128 DOTS_OCR_OVERLAP_DEC_CUS=128 $M --batch 64
run mixed_b32_queue $M --page-queue
