# flash attention v2 (4 waves x 64 rows): parity + A/B microbench
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4c; mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "flash or rope_split" 2>&1 | tail -15 ) > $O/pytest_flash.log 2>&1
( timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -k "flash" 2>&1 | tail -15 ) > $O/pytest_flash_full.log 2>&1
for m in 2 1 2 1; do
  ( DOTS_OCR_ATTN_MODE=$m timeout 200 python tools/microbench.py flash --seqs 8 --iters 6 ) 2>&1 | grep "flash attn" | sed "s/^/mode $m: /" >> $O/microbench_flash.txt
done
( DOTS_OCR_ATTN_MODE=2 timeout 200 python tools/microbench.py flash --seqs 2 --iters 6 ) 2>&1 | grep "flash attn" | sed "s/^/mode 2 (2 seqs): /" >> $O/microbench_flash.txt
tail -4 $O/pytest_flash.log; tail -4 $O/pytest_flash_full.log; cat $O/microbench_flash.txt
