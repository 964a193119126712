# GPU-box helper (edit per experiment).  Every command runs under its own short `timeout` and keeps its stderr:
# a silent crash followed by a hung profiler once cost 15 GPU-minutes.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 300 python $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err; echo rc=$?
tail -c 300 $R/gpurun_out/bench_default.json
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r7 -- python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 64 --no-cpu-baseline > $R/gpurun_out/prof_run7.log 2>&1; echo rc=$?
