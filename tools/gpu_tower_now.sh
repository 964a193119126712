# A/B: the next batch's tower launched beside this batch's prefill (DOTS_BENCH_TOWER_NOW=1) instead of behind it
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4n; mkdir -p $O
run() { name=$1; shift; ( timeout 400 "$@" ) > $O/$name.log 2>&1; grep '^{"metric"' $O/$name.log | tail -1 > $O/$name.json; }
DOTS_BENCH_TRACE=1 DOTS_BENCH_TOWER_NOW=1 DOTS_OCR_OVERLAP_DEC_CUS=64 run a4_64_64_now python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64
DOTS_BENCH_TRACE=1 DOTS_BENCH_TOWER_NOW=1 DOTS_OCR_OVERLAP_DEC_CUS=96 run a4_96_64_now python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64
DOTS_BENCH_TRACE=1 DOTS_BENCH_TOWER_NOW=1 DOTS_OCR_OVERLAP_DEC_CUS=128 run hr_128_64_now python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64
for f in a4_64_64_now a4_96_64_now hr_128_64_now; do grep "^\[step" $O/$f.log | tail -2; done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, d.get("parity_vs_sequential"), d.get("steps_checked"))
    except Exception as e: print(f, "FAILED", e); print(open(f.replace(".json",".log")).read()[-1500:])
PY
