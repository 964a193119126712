# NEXT-ROUND A/B (not run yet: the GPU budget of round 4 was spent): flash_attn64_kernel with the packed-fp32 exp stream (-DF64_PK).
# Before calling gpurun, on the build box:  bash tools/build_variant_attn.sh f64_pk "-DF64_PK"
# Then:  gpurun --timeout 600 -- 'bash tools/gpu_f64_pk.sh'
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/f64_pk; mkdir -p $O
V=$R/tools/bin/var_f64_pk/libdots_ocr_hip.so
# 1. parity of the variant: the flash-attention kernel tests and the real-shape tests through the variant library
DOTS_OCR_LIB=$V timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "attn or flash or vit or a4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
# 2. bit-identity with the default kernel is expected (same fused multiply-adds): same-box A/B, alternating
for i in 1 2; do
  ( echo -n "default: "; timeout 120 python tools/microbench.py flash --seqs 8 --iters 5 2>/dev/null | tail -1 ) >> $O/ab.txt
  ( echo -n "f64_pk:  "; DOTS_OCR_LIB=$V timeout 120 python tools/microbench.py flash --seqs 8 --iters 5 2>/dev/null | tail -1 ) >> $O/ab.txt
done
cat $O/ab.txt
