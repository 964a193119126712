# GPU-box helper: rocprofv3 --kernel-trace --stats over short bench runs -> gpurun_out/r03_*_kernel_stats.txt + the bench line of the SAME run
# (copied to profiles/ by hand).  Pass 1: --no-overlap (every kernel alone on the whole chip: comparable with rounds 1-2, agrees with
# roofline_sequential.avg_launch_ms).  Pass 2: the default command (software-pipelined batches: the tower's kernels run on a 128-CU partition).
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p $R/gpurun_out/prof
for mode in seq ovl; do
  rm -rf $R/gpurun_out/prof/*
  if [ $mode = seq ]; then FL="--no-overlap --max-new-tokens 64"; else FL="--max-new-tokens 128"; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r03 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $FL > $R/gpurun_out/prof/bench.log 2>&1; echo rc=$?
  grep '^{"metric"' $R/gpurun_out/prof/bench.log | tail -1 > $R/gpurun_out/r03_a4_b8_${mode}_bench_line.json
  db=$(find $R/gpurun_out/prof -name "*.db" | head -1); echo $db
  python profiles/summarize_rocprof.py $db > $R/gpurun_out/r03_a4_b8_${mode}_kernel_stats.txt; head -12 $R/gpurun_out/r03_a4_b8_${mode}_kernel_stats.txt
done
find $R/gpurun_out/prof -name "*.db" -size +20M -delete
