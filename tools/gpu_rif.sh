# rows-in-flight beyond 32: does a 48 / 64-row decode step on the small partition balance the tower?
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4k; mkdir -p $O
run() { name=$1; shift; ( timeout 400 "$@" ) > $O/$name.log 2>&1; grep '^{"metric"' $O/$name.log | tail -1 > $O/$name.json; }
DOTS_OCR_OVERLAP_DEC_CUS=64 run a4_dec64_rif48 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 48
DOTS_OCR_OVERLAP_DEC_CUS=64 run a4_dec64_rif64 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64
DOTS_OCR_OVERLAP_DEC_CUS=96 run a4_dec96_rif64 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64
DOTS_OCR_OVERLAP_DEC_CUS=128 run hr_dec128_rif32 python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 32
DOTS_OCR_OVERLAP_DEC_CUS=96 run hr_dec96_rif32 python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 32
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, d.get("parity_vs_sequential"), d.get("steps_checked"))
    except Exception as e: print(f, "FAILED", e); print(open(f.replace(".json",".log")).read()[-1500:])
PY
