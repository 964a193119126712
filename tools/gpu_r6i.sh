# Round 6, GPU call I: upper bound of an L2 prefetch of the small decode kernels' weights (VERDICT r5 #2b): B = 8 and B = 1 (fp8) with qkv / o_proj weights L2-resident
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6i; mkdir -p $O
db() { ( echo "== $1"; shift; env "$@" timeout 300 tools/bin/decode_bench $ARGS 2>&1 | grep -v amdgpu.ids ) >> $O/decode_bench.txt; }
ARGS="8 5700 6288"
db "8 rows, whole chip, distinct weights per layer (the real step)" X=1
db "8 rows, whole chip, qkv + o_proj weights L2-resident" DOTS_BENCH_SHARE_SMALL=1
db "8 rows, whole chip, all layer weights shared (Infinity Cache)" DOTS_BENCH_SHARE_ALL=1
ARGS="1 4500 4800"
db "1 row fp8, distinct weights" DOTS_BENCH_FP8=1
db "1 row fp8, qkv + o_proj weights L2-resident" DOTS_BENCH_FP8=1 DOTS_BENCH_SHARE_SMALL=1
db "1 row fp8, all layer weights shared" DOTS_BENCH_FP8=1 DOTS_BENCH_SHARE_ALL=1
grep -E "^==|whole step|^dec_qkv|^dec_proj o|^dec_gateup|^dec_proj down|^decode_attn" $O/decode_bench.txt
