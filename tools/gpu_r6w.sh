# Round 6, GPU call W: the driver's GPU test command on the final build (64-step anchor fixture), with durations
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6w; mkdir -p $O
T1=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=30 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$? wall=$(( $(date +%s) - T1 )) s"
grep -E "passed|failed|skipped" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-300
