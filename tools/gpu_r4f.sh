cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4f; mkdir -p $O
( timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py "tests/test_fullsize_vit_parity_gpu.py::test_fp8_engine_layer_by_layer_with_resynchronisation" -q -s 2>&1 | tail -40 ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log
