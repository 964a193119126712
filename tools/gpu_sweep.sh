# DEC_CUS sweep of the pipelined a4 bench (short runs)
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4b; mkdir -p $O
for c in ${SWEEP:-96 104 112 120}; do
  ( DOTS_OCR_OVERLAP_DEC_CUS=$c timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_dec$c.log 2>&1
  grep '^{"metric"' $O/bench_dec$c.log | tail -1 > $O/bench_dec$c.json
  python - $O/bench_dec$c.json $c <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); p=d['phase_ms_per_step']
    print('dec_cus',sys.argv[2],'pages/s %.3f step %.0f vit %.0f attn/launch %.2f prefill %.0f decode %.0f (%.3f ms/step)'%(d['value'],d['ms_per_step'],p['vit_ms'],d['roofline']['avg_launch_ms'],p['prefill_ms'],p['decode_ms'],d['roofline_decode']['ms_per_decode_step']), d.get('parity_vs_sequential'))
except Exception as e: print('fail',sys.argv[2],e)
PY
done
