# GPU-box helper (edit per experiment).  Every command runs under its own short `timeout` and keeps its stderr:
# a silent crash followed by a hung profiler once cost 15 GPU-minutes.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 300 python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench.out 2> $R/gpurun_out/bench.err; echo rc=$?
tail -c 400 $R/gpurun_out/bench.out; grep -v amdgpu.ids $R/gpurun_out/bench.err | tail -3
