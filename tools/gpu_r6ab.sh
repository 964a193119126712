# Round 6, GPU call AB: is the chip power-capped under the tower's kernels?  flash attention and the GEMMs ALONE on 256 / 192 / 128 / 64 CUs (CU-masked engine stream)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6ab; mkdir -p $O; rm -f $O/cus.txt
for rep in 1 2; do for hi in 255 191 127 63; do
  ( DOTS_OCR_CU_RANGE=0-$hi timeout 300 python tools/microbench.py flash gemm --seqs 8 --iters 4 ) 2>&1 | grep -E "flash attn|gemm " | sed "s/^/CUs $((hi+1)): /" >> $O/cus.txt
done; done
cat $O/cus.txt
