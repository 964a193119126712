cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err
tail -c 1500 $R/gpurun_out/bench_default.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r5 -- python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 64 --no-cpu-baseline > $R/gpurun_out/prof_run5.log 2>&1
