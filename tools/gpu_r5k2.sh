# Round 5, final session part 2: kernel stats (rocprofv3) of the sequential and the pipelined a4 command, PMC passes (GEMM counters + traffic,
# HBM traffic of the 64-row decode step on the partition), the bench lines of every workload on the final build.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5k; mkdir -p $O/prof
for mode in seq ovl; do
  rm -rf $O/prof/*
  if [ $mode = seq ]; then FL="--no-overlap --max-new-tokens 64"; else FL="--max-new-tokens 128"; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o r05 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $FL > $O/prof/bench.log 2>&1; echo "rocprof $mode rc=$?"
  grep '^{"metric"' $O/prof/bench.log | tail -1 > $O/r05_a4_b8_${mode}_bench_line.json
  db=$(find $O/prof -name "*.db" | head -1)
  python profiles/summarize_rocprof.py $db > $O/r05_a4_b8_${mode}_kernel_stats.txt; head -9 $O/r05_a4_b8_${mode}_kernel_stats.txt | cut -c1-150
done
rm -rf $O/prof
# ---- PMC: GEMM (one wave per SIMD) on the bench shapes
mkdir -p $O/pmcg
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/pmcg/p$i -- python $R/tools/gemm_bench.py > $O/pmcg/p$i.log 2>&1; echo "gemm pmc pass $i rc=$?"
done
python tools/pmc_summary.py $O/pmcg "gemm_bf16_w4_kernel" > $O/r05_pmc_gemm.json 2> $O/pmcg/summary.err; head -c 1500 $O/r05_pmc_gemm.json; rm -rf $O/pmcg
# ---- PMC: HBM-side traffic of the 64-row decode step on the 64-CU partition plan
mkdir -p $O/pmcd
CMD="$R/tools/bin/decode_bench 64 5700 6288 once"
DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcd/f -- $CMD > $O/pmcd/f.log 2>&1; echo "decode pmc fetch rc=$?"
DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcd/w -- $CMD > $O/pmcd/w.log 2>&1; echo "decode pmc write rc=$?"
python tools/pmc_summary.py $O/pmcd dec_qkv_wide decode_attn_kernel decode_attn_combine dec_proj_wide dec_gateup dec_lmhead dec_embed argmax > $O/r05_decode_traffic_64rows_raw.json 2> $O/pmcd/summary.err; cat $O/r05_decode_traffic_64rows_raw.json | head -c 2500; rm -rf $O/pmcd
# ---- bench lines
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "tail", d.get("overlap",{}).get("tower_tail_blocks"), "decode frac", d.get("roofline_decode",{}).get("frac"), "roofline", d.get("roofline",{}).get("frac"), d.get("parity_vs_sequential"), d.get("parity_vs_single_sequence"))
    except Exception as e: print(f, "FAILED", e)
PY
}
( timeout 300 python bench.py --rows-in-flight 8 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r05_bench_a4_one_batch_in_flight.json
cp $O/r05_bench_a4_one_batch_in_flight.json $R/profiles/r05_bench_a4_one_batch_in_flight.json
( timeout 700 python bench.py ) > $O/bench_a4.log 2>&1; grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/r05_bench_a4.json
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r05_bench_a4_torchrun_ws1.json
( timeout 300 python bench.py --workload highres --batch 4 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r05_bench_highres.json
( timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r05_bench_mixed64.json
( timeout 300 python bench.py --workload svg --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r05_bench_svg_fp8.json
( timeout 300 python bench.py --workload svg --fp8 0 --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r05_bench_svg_bf16.json
line $O/r05_bench_*.json
