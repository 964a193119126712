# Round 6, GPU call L: the heavy GPU tests after the shared-weights / gloo-loopback changes (durations), then the profiling pass (tools/gpu_r6_profiles.sh)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6l; mkdir -p $O
T1=$(date +%s)
timeout 1500 python -m pytest tests/test_multi_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_parity_gpu.py tests/test_fullsize_vit_parity_gpu.py tests/test_planted_walk_gpu.py tests/test_a4_anchor_gpu.py -x -q -m gpu --durations=12 > $O/pytest_heavy.log 2>&1; echo "heavy tests rc=$? wall=$(( $(date +%s) - T1 )) s"
grep -E "^[0-9.]+s (call|setup)|passed|failed" $O/pytest_heavy.log | head -16
bash tools/gpu_r6_profiles.sh
