# Round 5, GPU call C: the latency-batched epilogue of the one-wave-per-SIMD GEMM: parity + A/B
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c; mkdir -p $O
timeout 500 python -m pytest tests/test_gemm_plans_gpu.py -x -q -m gpu > $O/pytest_plans.log 2>&1; echo "plans pytest rc=$?"; tail -4 $O/pytest_plans.log
DOTS_OCR_GEMM_PLAN=1 timeout 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k gemm > $O/pytest_gemm_plan1.log 2>&1; echo "kernels(gemm, plan 1) rc=$?"; tail -3 $O/pytest_gemm_plan1.log
run() { ( echo "== $1"; shift; env "$@" timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_epi.txt; }
run "plan 0 (ping-pong)" DOTS_OCR_GEMM_PLAN=0
run "plan 1, batched epilogue" DOTS_OCR_GEMM_PLAN=1
run "plan 1, straight epilogue (-DW4_OLD_EPILOGUE)" DOTS_OCR_GEMM_PLAN=1 DOTS_OCR_LIB=$R/tools/bin/var_w4_oldepi/libdots_ocr_hip.so
run "plan 1, batched epilogue" DOTS_OCR_GEMM_PLAN=1
run "plan 1, no epilogue" DOTS_OCR_GEMM_PLAN=1 DOTS_OCR_LIB=$R/tools/bin/var_w4_nostore/libdots_ocr_hip.so
cat $O/gemm_epi.txt
