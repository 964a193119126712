cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
