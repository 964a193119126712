# GPU-box helper: HBM-side traffic of one decode step (VERDICT r1 next #2d).  Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
# over the standalone decode microbenchmark (B = 8, ctx 5700, 25 KV splits: the bench configuration), kernel trace only.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_dec; rm -rf $R/gpurun_out/pmc_dec/*
CMD="$R/tools/bin/decode_bench 8 5700 6288 once"
timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc_dec/f -- $CMD > $R/gpurun_out/pmc_dec/f.log 2>&1; echo rc=$?
timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc_dec/w -- $CMD > $R/gpurun_out/pmc_dec/w.log 2>&1; echo rc=$?
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_dec dec_qkv decode_attn_kernel decode_attn_combine dec_proj dec_gateup dec_lmhead dec_embed argmax > $R/gpurun_out/pmc_decode_summary.json 2> $R/gpurun_out/pmc_decode_summary.err; echo rc=$?
cat $R/gpurun_out/pmc_decode_summary.json; find $R/gpurun_out/pmc_dec -name "*.csv" -delete
