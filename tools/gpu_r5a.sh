# Round 5, GPU call A: the one-wave-per-SIMD GEMM (dots_set_gemm_plan(1)): parity + same-box A/B vs the ping-pong kernel + ablations;
# the -DF64_PK flash variant A/B (prepared in round 4, never run); decode baselines at 8 / 64 rows.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5a; mkdir -p $O
timeout 500 python -m pytest tests/test_gemm_plans_gpu.py -x -q -m gpu > $O/pytest_plans.log 2>&1; echo "plans pytest rc=$?"; tail -4 $O/pytest_plans.log
DOTS_OCR_GEMM_PLAN=1 timeout 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k gemm > $O/pytest_gemm_plan1.log 2>&1; echo "kernels(gemm, plan 1) rc=$?"; tail -3 $O/pytest_gemm_plan1.log
for i in 1 2; do
  ( echo "== plan 0 (ping-pong)"; DOTS_OCR_GEMM_PLAN=0 timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_ab.txt
  ( echo "== plan 1 (one wave per SIMD)"; DOTS_OCR_GEMM_PLAN=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_ab.txt
done
( echo "== plan 1, no epilogue (-DW4_NO_STORE)"; DOTS_OCR_LIB=$R/tools/bin/var_w4_nostore/libdots_ocr_hip.so DOTS_OCR_GEMM_PLAN=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_ab.txt
( echo "== plan 1, no DMA after the prologue, no epilogue"; DOTS_OCR_LIB=$R/tools/bin/var_w4_nodma/libdots_ocr_hip.so DOTS_OCR_GEMM_PLAN=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_ab.txt
( echo "== plan 0, no epilogue (-DPP_NO_STORE)"; DOTS_OCR_LIB=$R/tools/bin/var_pp_nostore/libdots_ocr_hip.so DOTS_OCR_GEMM_PLAN=0 timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_ab.txt
cat $O/gemm_ab.txt
# ---- flash_attn64 -DF64_PK
V=$R/tools/bin/var_f64_pk/libdots_ocr_hip.so
DOTS_OCR_LIB=$V timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn" > $O/pytest_f64pk.log 2>&1; echo "f64_pk pytest rc=$?"; tail -3 $O/pytest_f64pk.log
for i in 1 2; do
  ( echo -n "default: "; timeout 120 python tools/microbench.py flash --seqs 8 --iters 5 2>/dev/null | tail -1 ) >> $O/f64pk_ab.txt
  ( echo -n "f64_pk:  "; DOTS_OCR_LIB=$V timeout 120 python tools/microbench.py flash --seqs 8 --iters 5 2>/dev/null | tail -1 ) >> $O/f64pk_ab.txt
done
cat $O/f64pk_ab.txt
# ---- decode baselines
( timeout 120 tools/bin/decode_bench 8 5700 6288 ) > $O/decode_b8.txt 2>&1; tail -25 $O/decode_b8.txt
( timeout 120 tools/bin/decode_bench 64 5700 6288 ) > $O/decode_b64.txt 2>&1; tail -25 $O/decode_b64.txt
( DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 120 tools/bin/decode_bench 64 5700 6288 ) > $O/decode_b64_64cus.txt 2>&1; tail -25 $O/decode_b64_64cus.txt
