# final lines of the round on the two-tile build: remaining model tests + smoke, kernel stats of the pipelined command, default bench, torchrun, highres, mixed64
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4r; mkdir -p $O/prof
timeout 400 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "not (batch or continuous or server or slots)" > $O/pytest_model_rest.log 2>&1; echo "pytest model rest rc=$?"; tail -2 $O/pytest_model_rest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
rm -rf $O/prof/*
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o r04 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 128 > $O/prof/bench.log 2>&1; echo "rocprof ovl rc=$?"
grep '^{"metric"' $O/prof/bench.log | tail -1 > $O/r04_a4_b8_ovl_bench_line.json
db=$(find $O/prof -name "*.db" | head -1)
python profiles/summarize_rocprof.py $db > $O/r04_a4_b8_ovl_kernel_stats.txt; head -5 $O/r04_a4_b8_ovl_kernel_stats.txt
find $O/prof -name "*.db" -size +20M -delete
( timeout 600 python bench.py ) > $O/bench_a4.log 2>&1; grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/r04_bench_a4.json
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_a4_torchrun_ws1.json
( timeout 300 python bench.py --workload highres --batch 4 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_highres.json
( timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_mixed64.json
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/r04_bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "dec frac", d.get("roofline_decode",{}).get("frac"), d.get("roofline",{}).get("traffic"), d.get("parity_vs_sequential"), d.get("steps_checked"))
    except Exception as e: print(f, "FAILED", e)
PY
