# Round 6, GPU call AF: everything on the fused-qkv build: the driver's GPU test command, smoke, kernel stats (sequential / pipelined), the driver's bench command
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R
bash tools/gpu_r6w.sh
bash tools/gpu_r6_final2.sh
