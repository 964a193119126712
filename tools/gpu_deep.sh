cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4g; mkdir -p $O
for cfgs in ${CFGS:-"64 24" "64 32" "96 24" "96 32" "32 32"}; do
  set -- $cfgs; c=$1; r=$2
  ( DOTS_OCR_OVERLAP_DEC_CUS=$c timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rows-in-flight $r ) > $O/bench_dec${c}_rif$r.log 2>&1
  grep '^{"metric"' $O/bench_dec${c}_rif$r.log | tail -1 > $O/bench_dec${c}_rif$r.json
  python - $O/bench_dec${c}_rif$r.json $c $r <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); p=d['phase_ms_per_step']
    print('dec_cus',sys.argv[2],'rif',sys.argv[3],'pages/s %.3f step %.0f vit %.0f attn/launch %.2f prefill %.0f decode %.0f (%.3f ms/step)'%(d['value'],d['ms_per_step'],p['vit_ms'],d['roofline']['avg_launch_ms'],p['prefill_ms'],p['decode_ms'],d['roofline_decode']['ms_per_decode_step']), d.get('parity_vs_sequential'), d.get('steps_checked'), 'dec frac', round(d['roofline_decode']['frac'],3))
except Exception as e:
    print('fail',sys.argv[2],sys.argv[3],e); print(open(sys.argv[1].replace('.json','.log')).read()[-1500:])
PY
done
