# Round 6, profiling call: rocprofv3 kernel stats of the sequential and the pipelined a4 command, HBM-side traffic (PMC, separate FETCH / WRITE passes)
# of the 64-row decode step on the 64-CU partition and of the ViT flash-attention kernel.  Summaries land in gpurun_out/r6p/ (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6p; mkdir -p $O/prof
for mode in seq ovl; do
  rm -rf $O/prof/*
  if [ $mode = seq ]; then FL="--no-overlap --max-new-tokens 64"; else FL="--max-new-tokens 128"; fi
  DOTS_BENCH_OTHER=0 timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o r06 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $FL > $O/prof/bench.log 2>&1; echo "rocprof $mode rc=$?"
  grep '^{"metric"' $O/prof/bench.log | tail -1 > $O/r06_a4_b8_${mode}_bench_line.json
  db=$(find $O/prof -name "*.db" | head -1)
  python profiles/summarize_rocprof.py $db > $O/r06_a4_b8_${mode}_kernel_stats.txt; head -12 $O/r06_a4_b8_${mode}_kernel_stats.txt | cut -c1-150
done
rm -rf $O/prof
# ---- PMC: HBM-side traffic of the 64-row decode step on the 64-CU partition plan
mkdir -p $O/pmcd
CMD="$R/tools/bin/decode_bench 64 5700 6288 once"
DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcd/f -- $CMD > $O/pmcd/f.log 2>&1; echo "decode pmc fetch rc=$?"
DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcd/w -- $CMD > $O/pmcd/w.log 2>&1; echo "decode pmc write rc=$?"
python tools/pmc_summary.py $O/pmcd dec_qkv_wide decode_attn_kernel decode_attn_combine dec_proj_wide dec_proj_ksplit "dec_stream64_kernel<0" "dec_stream64_kernel<1" dec_norm_ximg dec_embed argmax > $O/r06_decode_traffic_64rows_raw.json 2> $O/pmcd/summary.err; head -c 3000 $O/r06_decode_traffic_64rows_raw.json; rm -rf $O/pmcd
# ---- PMC: HBM-side traffic of the ViT flash attention (one sequential batch, 2 new tokens)
mkdir -p $O/pmcf
CMD="python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 2 --no-cpu-baseline --no-overlap"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcf/f -- $CMD > $O/pmcf/f.log 2>&1; echo "flash pmc fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcf/w -- $CMD > $O/pmcf/w.log 2>&1; echo "flash pmc write rc=$?"
python tools/pmc_summary.py $O/pmcf "flash_attn64_kernel<false" "flash_attn64_kernel<true" "gemm_bf16_w4_kernel" > $O/r06_flash_attn_traffic_raw.json 2> $O/pmcf/summary.err; head -c 2000 $O/r06_flash_attn_traffic_raw.json; rm -rf $O/pmcf
