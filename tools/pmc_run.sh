cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
python -m pytest $R/tests/test_model_gpu.py -x -q 2>&1 | tail -2
python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --max-new-tokens 256 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['phase_ms_per_step']['decode_ms'], d['roofline_decode']['ms_per_decode_step'])"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r4 -- python $R/bench.py --steps 1 --warmup 0 --max-new-tokens 64 --no-cpu-baseline > $R/gpurun_out/prof_run4.log 2>&1
