# GPU-box helper (edit per experiment).  Every command runs under its own short `timeout` and keeps its stderr:
# a silent crash followed by a hung profiler once cost 15 GPU-minutes.  --pmc runs use --kernel-trace only.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc2; rm -rf $R/gpurun_out/pmc2/*
CMD="python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 2 --no-cpu-baseline --no-overlap"
timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc2/f -- $CMD > $R/gpurun_out/pmc2/f.log 2>&1; echo rc=$?
timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc2/w -- $CMD > $R/gpurun_out/pmc2/w.log 2>&1; echo rc=$?
python $R/tools/pmc_summary.py $R/gpurun_out/pmc2 "flash_attn_kernel<false" "gemm_bf16_256_kernel" > $R/gpurun_out/pmc2_summary.json 2> $R/gpurun_out/pmc2_summary.err; echo rc=$?
cat $R/gpurun_out/pmc2_summary.json; find $R/gpurun_out/pmc2 -name "*.csv" -delete
