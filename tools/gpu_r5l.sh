# Round 5, GPU call L: mixed64 with the cold-start ramp (admit_group) + tower readiness gate: admission group / look-ahead sizes and decode partitions.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5l; mkdir -p $O
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "towers", d.get("roofline",{}).get("towers"), "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_single_sequence"))
    except Exception as e: print(f, "FAILED", e)
PY
}
run() { n=$1; shift; ( env "$@" timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/$n.log 2>&1; grep '^{"metric"' $O/$n.log | tail -1 > $O/$n.json; line $O/$n.json; }
run m64_default
run m64_g8_cus96 DOTS_BENCH_PREFETCH=8 DOTS_BENCH_ADMIT_GROUP=8
run m64_g8_cus64 DOTS_BENCH_PREFETCH=8 DOTS_BENCH_ADMIT_GROUP=8 DOTS_OCR_OVERLAP_DEC_CUS=64
run m64_g16_cus96 DOTS_BENCH_PREFETCH=16 DOTS_BENCH_ADMIT_GROUP=16
run m64_g4_cus64 DOTS_BENCH_PREFETCH=4 DOTS_BENCH_ADMIT_GROUP=4 DOTS_OCR_OVERLAP_DEC_CUS=64
tail -3 $O/m64_g8_cus96.log | cut -c1-300
