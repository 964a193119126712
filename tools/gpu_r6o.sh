# Round 6, GPU call O: the projections above 32 rows as four K quarters: bitwise tests (every MFMA count, o_proj too), decode_bench A/B of DOTS_OCR_DEC_KSPLIT = 0 / 1 / 2
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6o; mkdir -p $O
timeout 1500 python -m pytest tests/test_decode_kernels_gpu.py tests/test_decode_plans_gpu.py tests/test_fp8_gpu.py -x -q -m gpu > $O/pytest_decode.log 2>&1; echo "decode kernels + plans + fp8 pytest rc=$?"; tail -3 $O/pytest_decode.log
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="64 5700 6288"
for m in 0 1 2; do
  db "64 rows, 64-CU partition plan, DOTS_OCR_DEC_KSPLIT=$m" DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_KSPLIT=$m
  db "64 rows, whole chip, DOTS_OCR_DEC_KSPLIT=$m" DOTS_OCR_DEC_KSPLIT=$m
  db "64 rows, 128-CU partition plan, DOTS_OCR_DEC_KSPLIT=$m" DOTS_BENCH_CUS=128 DOTS_BENCH_FULL=1 DOTS_OCR_DEC_KSPLIT=$m
done
grep -E "^==|whole step|dec_proj down  |dec_proj o   |dec_qkv  |dec_gateup  |marginal dec_proj|marginal dec_qkv|marginal dec_gateup" $O/decode_bench.txt
