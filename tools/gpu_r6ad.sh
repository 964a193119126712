# Round 6, GPU call AD: the qkv GEMM's rope epilogue (q / k leave the GEMM rotated and head-major): bitwise test against GEMM + split, then same-box A/B of the a4 bench
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6ad; mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rope or gemm" ) > $O/pytest.log 2>&1; echo "kernel tests rc=$?"; tail -12 $O/pytest.log | cut -c1-300
for rep in 1 2; do for f in 0 1; do
( DOTS_OCR_QK_FUSE=$f DOTS_BENCH_DECODE_ALONE=0 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs ) > $O/b_$f.log 2>&1; grep '^{"metric"' $O/b_$f.log | tail -1 > $O/b_$f.json
python - $O/b_$f.json $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); s=d["throughput_shapes"]["sequential_batch"]
    print("fuse", sys.argv[2], "a4 %.3f pages/s step %.1f"%(d["value"], d["ms_per_step"]), {k:round(v,1) for k,v in d["phase_ms_per_step"].items() if k in ("vit_ms","prefill_ms","decode_ms","vit_attn_ms")}, "seq vit frac %.4f"%d["roofline_vit_sequential"]["frac"], d.get("parity_vs_sequential"))
except Exception as e: print("fuse", sys.argv[2], "FAILED", e)
PY
done; done
