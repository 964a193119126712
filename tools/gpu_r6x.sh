# Round 6, GPU call X: decode attention waves-per-workgroup (= KV split) A/B at 64 and 8 rows: is a finer split faster above 32 rows?
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6x; mkdir -p $O; rm -f $O/decode_bench.txt
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
for wv in 4 8; do
ARGS="64 5700 6288"
db "64 rows, whole chip, attention waves $wv" DOTS_OCR_ATTN_WAVES=$wv
db "64 rows, 64-CU partition plan, attention waves $wv" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_ATTN_WAVES=$wv
ARGS="8 5700 6288"
db "8 rows, whole chip, attention waves $wv" DOTS_OCR_ATTN_WAVES=$wv
done
grep -E "^==|whole step|^decode_attn " $O/decode_bench.txt
( DOTS_OCR_ATTN_WAVES=8 timeout 900 python -m pytest tests/test_decode_kernels_gpu.py -x -q -m gpu -k "attention" ) > $O/pytest_w8.log 2>&1; echo "attention tests with 8 waves rc=$?"; tail -2 $O/pytest_w8.log
