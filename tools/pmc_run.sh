cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fullsize_gpu.py -x -q -k full_size 2>&1 | tail -3
