# host-side timeline of the multi-batch pipeline's steps (DOTS_BENCH_TRACE=1): where does a step's time outside tower / prefill / decode go?
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4m; mkdir -p $O
( DOTS_BENCH_TRACE=1 DOTS_OCR_OVERLAP_DEC_CUS=64 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 64 ) > $O/a4_64_64.log 2>&1
( DOTS_BENCH_TRACE=1 DOTS_OCR_OVERLAP_DEC_CUS=64 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight 48 ) > $O/a4_64_48.log 2>&1
grep "^\[step" $O/a4_64_64.log | tail -6; grep "^\[step" $O/a4_64_48.log | tail -4
