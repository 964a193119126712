# Round 6, GPU call H: GEMM epilogue swizzle A/B (same box, interleaved), GEMM tests, PMC bank-conflict counters for both builds
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6h; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_plans_gpu.py -x -q -m gpu -k "gemm" > $O/pytest_gemm.log 2>&1; echo "gemm pytest rc=$?"; tail -2 $O/pytest_gemm.log
for rep in 1 2 3; do
  echo "== new swizzle (rep $rep)" >> $O/gemm_ab.txt; timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm_ab.txt
  echo "== round-5 swizzle (rep $rep)" >> $O/gemm_ab.txt; DOTS_OCR_LIB=$R/tools/bin/var_old_swz/libdots_ocr_hip.so timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm_ab.txt
done
grep -E "^==|weighted" $O/gemm_ab.txt
mkdir -p $O/pmc
for v in new old; do
  if [ $v = old ]; then export DOTS_OCR_LIB=$R/tools/bin/var_old_swz/libdots_ocr_hip.so; else unset DOTS_OCR_LIB; fi
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc/$v -- python $R/tools/gemm_bench.py > $O/pmc/$v.log 2>&1; echo "pmc $v rc=$?"
  python tools/pmc_summary.py $O/pmc/$v "gemm_bf16_w4_kernel" > $O/pmc_gemm_$v.json 2>/dev/null; python - $O/pmc_gemm_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items(): print(sys.argv[1].split('/')[-1], {c:round(x["mean_per_dispatch"]) for c,x in v.items()})
PY
done
rm -rf $O/pmc
