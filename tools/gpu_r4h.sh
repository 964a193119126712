cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4h; mkdir -p $O/pmc; rm -rf $O/pmc/*
# ---- SQ counters + HBM traffic of flash_attn64_kernel (separate --pmc passes, kernel trace only)
CMD="python $R/tools/microbench.py flash --seqs 8 --iters 2"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/pmc/p$i -- $CMD > $O/pmc/p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python $R/tools/pmc_summary.py $O/pmc "flash_attn64_kernel" > $O/pmc_flash64.json 2> $O/pmc/summary.err; echo rc=$?
python - $O/pmc <<'PY'
import csv,glob,sys,collections
# average duration of the kernel in the traced passes
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "flash_attn64" in row.get("Kernel_Name",""):
            d[f.split("/pmc/")[1].split("/")[0]].append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e6)
for k,v in sorted(d.items()): print("pass",k,"launches",len(v),"avg ms %.3f"%(sum(v)/len(v)))
PY
find $O/pmc -name "*.csv" -delete
cat $O/pmc_flash64.json | head -60
# ---- other configurations on this build
( timeout 300 python bench.py --workload svg --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_svg_fp8.json
( timeout 300 python bench.py --workload svg --fp8 0 --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_svg_bf16.json
( timeout 300 python bench.py --workload highres --batch 4 --steps 3 --warmup 1 --no-cpu-baseline ) > $O/bench_highres.log 2>&1; grep '^{"metric"' $O/bench_highres.log | tail -1 > $O/bench_highres.json
for pf in 0 8 32; do
  ( DOTS_BENCH_PREFETCH=$pf DOTS_OCR_OVERLAP_DEC_CUS=96 timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_mixed64_pf$pf.log 2>&1; grep '^{"metric"' $O/bench_mixed64_pf$pf.log | tail -1 > $O/bench_mixed64_pf$pf.json
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "dec frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"))
    except Exception as e: print(f, "FAILED", e)
PY
