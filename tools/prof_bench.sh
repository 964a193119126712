# GPU-box helper: rocprofv3 --kernel-trace --stats over a short bench run -> gpurun_out/r02_kernel_stats.txt (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof; rm -rf $R/gpurun_out/prof/*
cd $R
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 0 --max-new-tokens 64 --no-cpu-baseline > $R/gpurun_out/prof/bench.log 2>&1; echo rc=$?
tail -1 $R/gpurun_out/prof/bench.log > $R/gpurun_out/r02_prof_bench_line.json
db=$(find $R/gpurun_out/prof -name "*.db" | head -1); echo $db
python profiles/summarize_rocprof.py $db > $R/gpurun_out/r02_kernel_stats.txt; head -30 $R/gpurun_out/r02_kernel_stats.txt
find $R/gpurun_out/prof -name "*.db" -size +20M -delete
