# Round 6, GPU call R: HBM-side traffic of the 64-row decode step on the final build (the K-split down_proj included), highres leg with its own default partition
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6p; mkdir -p $O/pmcd
CMD="$R/tools/bin/decode_bench 64 5700 6288 once"
DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcd/f -- $CMD > $O/pmcd/f.log 2>&1; echo "decode pmc fetch rc=$?"
DOTS_BENCH_CUS=64 DOTS_BENCH_FULL=1 timeout 170 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcd/w -- $CMD > $O/pmcd/w.log 2>&1; echo "decode pmc write rc=$?"
python tools/pmc_summary.py $O/pmcd dec_qkv_wide decode_attn_kernel decode_attn_combine dec_proj_wide dec_proj_ksplit "dec_stream64_kernel<0" "dec_stream64_kernel<1" dec_norm_ximg dec_embed argmax > $O/r06_decode_traffic_64rows_raw.json 2> $O/pmcd/summary.err; rm -rf $O/pmcd
python - $O/r06_decode_traffic_64rows_raw.json <<'PY'
import json,sys
raw=json.load(open(sys.argv[1])); tot=0
for k,v in raw.items():
    f=v.get("FETCH_SIZE",{}).get("mean_per_dispatch",0); w=v.get("WRITE_SIZE",{}).get("mean_per_dispatch",0); n=v.get("FETCH_SIZE",{}).get("dispatches",0)
    tot+=(2*f+w)*1024*n/3.0; print("%-26s x%5.1f/step %8.1f MB/launch"%(k,n/3.0,(2*f+w)*1024/1e6))
print("total %.3f GB per step"%(tot/1e9))
PY
( DOTS_BENCH_OTHER=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > $O/bench_default_short.log 2>&1; grep '^{"metric"' $O/bench_default_short.log | tail -1 > $O/bench_default_short.json
python - $O/bench_default_short.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("a4", round(d["value"],3))
for k,v in d.get("other_configs",{}).items(): print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","failed","leg_wall_s")})
PY
