# GPU-box helper (edit per experiment).  Every command runs under its own short `timeout` and keeps its stderr:
# a silent crash followed by a hung profiler once cost 15 GPU-minutes.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f3 -- python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 2 --no-cpu-baseline > $R/gpurun_out/pmc_fetch3.log 2>&1; echo rc=$?
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w3 -- python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 2 --no-cpu-baseline > $R/gpurun_out/pmc_write3.log 2>&1; echo rc=$?
