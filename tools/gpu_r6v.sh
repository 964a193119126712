# Round 6, GPU call V: the 64-step anchor fixture: anchor test + the a4 decode parity test (now without its inline oracle)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6v; mkdir -p $O
( timeout 900 python -m pytest tests/test_a4_anchor_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu --durations=5 -s ) > $O/pytest.log 2>&1; echo "rc=$?"; grep -v "^$" $O/pytest.log | tail -25 | cut -c1-300
