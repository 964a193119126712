# Round 5, GPU call B: why does the GEMM epilogue cost 7-21 us per tile?  (1) staggered first round, (2) the same kernels on 64 CUs
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5b; mkdir -p $O
export DOTS_OCR_GEMM_PLAN=1
run() { ( echo "== $1"; shift; env "$@" timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids ) >> $O/gemm_stagger.txt; }
run "plan 1" X=1
run "plan 1, staggered start (1/8 tile per phase)" DOTS_OCR_LIB=$R/tools/bin/var_w4_stagger/libdots_ocr_hip.so
run "plan 1, staggered start (2/8 tile per phase)" DOTS_OCR_LIB=$R/tools/bin/var_w4_stagger14/libdots_ocr_hip.so
run "plan 1" X=1
run "plan 1 on 64 CUs (ROC_GLOBAL_CU_MASK)" ROC_GLOBAL_CU_MASK=0xffffffffffffffff
run "plan 1 no epilogue on 64 CUs" ROC_GLOBAL_CU_MASK=0xffffffffffffffff DOTS_OCR_LIB=$R/tools/bin/var_w4_nostore/libdots_ocr_hip.so
cat $O/gemm_stagger.txt
