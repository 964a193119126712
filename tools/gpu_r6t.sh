# Round 6, GPU call T: flash_attn64_kernel with the row maxima rescheduled (-DF64_TAILMAX): parity tests on the variant, same-box A/B
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6t; mkdir -p $O; rm -f $O/ab.txt
V=$R/tools/bin/var_tm_lr/libdots_ocr_hip.so
( DOTS_OCR_LIB=$V timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "flash or attn" ) > $O/pytest_variant.log 2>&1; echo "variant kernel tests rc=$?"; tail -2 $O/pytest_variant.log
for rep in 1 2; do for v in base late_read tm_lr tailmax; do
  if [ $v = base ]; then L=""; else L="DOTS_OCR_LIB=$R/tools/bin/var_$v/libdots_ocr_hip.so"; fi
  ( env $L timeout 200 python tools/microbench.py flash --seqs 8 --iters 6 ) 2>&1 | grep "flash attn" | sed "s/^/$v: /" >> $O/ab.txt
done; done
cat $O/ab.txt
