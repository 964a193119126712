# GPU-box helper: SQ counters + HBM traffic of the round-2 GEMM on the bench's shapes (tools/gemm_bench.py), separate --pmc passes,
# kernel trace only.  Summary -> gpurun_out/r02_pmc_gemm.json
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmcg; rm -rf $R/gpurun_out/pmcg/*
CMD="python $R/tools/gemm_bench.py"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/gpurun_out/pmcg/p$i -- $CMD > $R/gpurun_out/pmcg/p$i.log 2>&1; echo "pass $i rc=$?"
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmcg "gemm_bf16_256pp_kernel" > $R/gpurun_out/r02_pmc_gemm.json 2> $R/gpurun_out/pmcg/summary.err; echo rc=$?
cat $R/gpurun_out/r02_pmc_gemm.json; find $R/gpurun_out/pmcg -name "*.csv" -delete
