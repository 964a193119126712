cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -x -q -k sampling 2>&1 | tail -12
