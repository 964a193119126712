cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 1 --warmup 0 --max-new-tokens 16 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-600
