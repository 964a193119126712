# Round 5, GPU call O: decode-partition size re-checked for highres and mixed64 now that the wide kernels made the decode step cheaper.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5o; mkdir -p $O
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.4f %s ms/step %.1f"%(d["value"],d["unit"],d["ms_per_step"]), {k:round(v,1) for k,v in d.get("phase_ms_per_step",{}).items()}, "decode frac", d.get("roofline_decode",{}).get("frac"), d.get("parity_vs_sequential"), d.get("parity_vs_single_sequence"))
    except Exception as e: print(f, "FAILED", e)
PY
}
for c in 128 96 64; do
  ( DOTS_OCR_OVERLAP_DEC_CUS=$c timeout 300 python bench.py --workload highres --batch 4 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/highres_cus$c.json; line $O/highres_cus$c.json
done
for c in 96 64; do
  ( DOTS_OCR_OVERLAP_DEC_CUS=$c timeout 300 python bench.py --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline ) 2>&1 | grep '^{"metric"' | tail -1 > $O/mixed64_cus$c.json; line $O/mixed64_cus$c.json
done
