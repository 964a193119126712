# time-sliced multi-batch pipeline (no CU partitions) against the partitioned default, rows in flight 32 / 64; highres at 64 rows
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4l; mkdir -p $O
run() { name=$1; shift; ( timeout 400 "$@" ) > $O/$name.log 2>&1; grep '^{"metric"' $O/$name.log | tail -1 > $O/$name.json; }
run a4_sliced_rif64 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64 --time-sliced
run a4_sliced_rif32 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 32 --time-sliced
run hr_sliced_rif32 python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 32 --time-sliced
run hr_sliced_rif64 python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64 --time-sliced
DOTS_OCR_OVERLAP_DEC_CUS=128 run hr_dec128_rif64 python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "dec", round(d["roofline_decode"]["frac"],4), round(d["roofline_decode"]["ms_per_decode_step"],3), "attn", round(d["roofline"]["frac"],4), d.get("parity_vs_sequential"), d.get("steps_checked"))
    except Exception as e: print(f, "FAILED", e); print(open(f.replace(".json",".log")).read()[-1500:])
PY
