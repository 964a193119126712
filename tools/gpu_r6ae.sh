# Round 6, GPU call AE: the rope epilogue after pinning the second fma contraction: bitwise test, model / fullsize / overlap tests on the fused path
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6ae; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_overlap_gpu.py tests/test_a4_anchor_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.log | cut -c1-300
