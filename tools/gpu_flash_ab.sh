# flash attention v2 ablations (timing only: the ablated builds compute garbage)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4d; mkdir -p $O; rm -f $O/ab.txt
for v in base ${VARIANTS}; do
  if [ $v = base ]; then L=""; else L="DOTS_OCR_LIB=$R/tools/bin/var_$v/libdots_ocr_hip.so"; fi
  ( env $L DOTS_OCR_ATTN_MODE=2 timeout 200 python tools/microbench.py flash --seqs 8 --iters 6 ) 2>&1 | grep "flash attn" | sed "s/^/$v: /" >> $O/ab.txt
done
( DOTS_OCR_ATTN_MODE=1 timeout 200 python tools/microbench.py flash --seqs 8 --iters 6 ) 2>&1 | grep "flash attn" | sed "s/^/mode1: /" >> $O/ab.txt
cat $O/ab.txt
