# Round-4 final measurement run: rocprofv3 kernel stats (sequential + pipelined), decode PMC traffic, bench lines of every configuration
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4j; mkdir -p $O/prof
for mode in seq ovl; do
  rm -rf $O/prof/*
  if [ $mode = seq ]; then FL="--no-overlap --max-new-tokens 64"; else FL="--max-new-tokens 128"; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o r04 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $FL > $O/prof/bench.log 2>&1; echo "rocprof $mode rc=$?"
  grep '^{"metric"' $O/prof/bench.log | tail -1 > $O/r04_a4_b8_${mode}_bench_line.json
  db=$(find $O/prof -name "*.db" | head -1)
  python profiles/summarize_rocprof.py $db > $O/r04_a4_b8_${mode}_kernel_stats.txt; head -8 $O/r04_a4_b8_${mode}_kernel_stats.txt
done
find $O/prof -name "*.db" -size +20M -delete
# decode traffic (B = 8, ctx 5700): FETCH_SIZE / WRITE_SIZE passes over the standalone decode microbenchmark
mkdir -p $O/pmc_dec; rm -rf $O/pmc_dec/*
CMD="$R/tools/bin/decode_bench 8 5700 6288 once"
timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_dec/f -- $CMD > $O/pmc_dec/f.log 2>&1; echo rc=$?
timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_dec/w -- $CMD > $O/pmc_dec/w.log 2>&1; echo rc=$?
python $R/tools/pmc_summary.py $O/pmc_dec dec_qkv decode_attn_kernel decode_attn_combine dec_proj dec_gateup dec_lmhead dec_embed argmax > $O/pmc_decode_summary.json 2> $O/pmc_dec/summary.err; echo rc=$?
find $O/pmc_dec -name "*.csv" -delete
( timeout 120 tools/bin/decode_bench 8 5700 6288 ) > $O/decode_bench_whole.txt 2>&1
# bench lines
( timeout 600 python bench.py ) > $O/bench_a4.log 2>&1; grep '^{"metric"' $O/bench_a4.log | tail -1 > $O/r04_bench_a4.json
( timeout 300 python bench.py --no-cpu-baseline --rows-in-flight 8 ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_a4_one_batch_decoding.json
( timeout 300 python bench.py --no-cpu-baseline --no-overlap ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_a4_sequential.json
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_a4_torchrun_ws1.json
( timeout 300 python bench.py --workload highres --batch 4 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_highres.json
( timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_mixed64.json
( DOTS_BENCH_PREFETCH=0 timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_mixed64_no_lookahead.json
( timeout 300 python bench.py --workload svg --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_svg_fp8.json
( timeout 300 python bench.py --workload svg --fp8 0 --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_svg_bf16.json
( timeout 300 python bench.py --workload svg --batch 8 --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_svg_fp8_b8.json
( timeout 300 python bench.py --workload svg --batch 16 --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/r04_bench_svg_fp8_b16.json
( timeout 400 python tools/serve_bench.py --pages 32 --slots 8 ) 2>&1 | tail -1 > $O/r04_serve_bench_a4_8slots.json
( timeout 400 python tools/serve_bench.py --pages 48 --slots 16 ) 2>&1 | tail -1 > $O/r04_serve_bench_a4_16slots.json
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/r04_bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "dec frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"), d.get("steps_checked"))
    except Exception as e: print(f, "FAILED", e)
for f in sorted(glob.glob(sys.argv[1]+"/r04_serve*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], {k:(v["pages_per_s"] if isinstance(v,dict) else v) for k,v in d.items() if k in ("static","continuous","continuous_prefetch1","continuous_prefetch2","continuous_prefetch4","continuous_prefetch8","identical_tokens")})
    except Exception as e: print(f, "FAILED", e)
PY
grep "whole step\|^dec_\|^decode_attn" $O/decode_bench_whole.txt
