cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 120 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 8 > $R/gpurun_out/dbg.out 2> $R/gpurun_out/dbg.err; echo rc=$?
tail -c 300 $R/gpurun_out/dbg.out; grep -v amdgpu.ids $R/gpurun_out/dbg.err | tail -5
