# Round 5, final session part 1: the whole -m gpu suite on the final tree (what the driver runs at round end) with durations, then smoke().
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5k; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=30 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -45 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
