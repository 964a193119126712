cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2
python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_a4_v7.json 2> $R/gpurun_out/bench_a4_v7.err
python -c "
import json; d=json.load(open('$R/gpurun_out/bench_a4_v7.json')); print(d['value'], d['phase_ms_per_step'], d['roofline']['achieved'], d['roofline_vit']['achieved'], d['roofline_decode']['ms_per_decode_step'])"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f2 -- python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 2 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
tail -1 $R/gpurun_out/pmc_fetch.log
