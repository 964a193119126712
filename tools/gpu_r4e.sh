cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4e; mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5 ) > $O/pytest_kernels.log 2>&1
tail -2 $O/pytest_kernels.log
SWEEP="96 128" bash tools/gpu_sweep.sh
