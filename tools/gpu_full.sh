cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4i; mkdir -p $O
( timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -30 ) > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -2
