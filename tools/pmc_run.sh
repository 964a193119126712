# GPU-box helper (edit per experiment).  Every command runs under its own short `timeout` and keeps its stderr:
# a silent crash followed by a hung profiler once cost 15 GPU-minutes.  --pmc runs use --kernel-trace only.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
( cd $R && timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
timeout 300 python $R/bench.py > $R/gpurun_out/bench_final2.json 2> $R/gpurun_out/bench_final2.err; echo rc=$?
tail -c 1800 $R/gpurun_out/bench_final2.json
DOTS_OCR_ATTN_MODE=0 timeout 200 python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_mode0.json 2> $R/gpurun_out/bench_mode0.err; echo rc=$?
tail -c 700 $R/gpurun_out/bench_mode0.json
rm -rf $R/gpurun_out/prof2
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2 -o r8 -- python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 64 --no-cpu-baseline > $R/gpurun_out/prof_run8.log 2>&1; echo rc=$?
ls -la $R/gpurun_out/prof2 | head
