# Round 6, GPU call F: in-kernel timeline (DOTS_TRACE build) of the 64-row step on the 64-CU partition and on the whole chip
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6f; mkdir -p $O
( echo "== 64 rows, 64-CU partition plan (trace build)"; DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 300 tools/bin/decode_bench_trace 64 5700 6288 2>&1 | grep -v amdgpu.ids ) > $O/trace_b64_partition.txt
( echo "== 64 rows, whole chip (trace build)"; timeout 300 tools/bin/decode_bench_trace 64 5700 6288 2>&1 | grep -v amdgpu.ids ) > $O/trace_b64_chip.txt
grep -E "^==|whole step|^dec_|trace dec_" $O/trace_b64_partition.txt $O/trace_b64_chip.txt | cut -c1-400
