# GPU-box helper (edit per experiment).  Every command runs under its own short `timeout` and keeps its stderr:
# a silent crash followed by a hung profiler once cost 15 GPU-minutes.
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 300 python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench.out 2> $R/gpurun_out/bench.err; echo rc=$?
python -c "
import json; d=json.load(open('$R/gpurun_out/bench.out')); print(d['value'], d['phase_ms_per_step'], d['roofline']['achieved'], d['roofline_vit']['achieved'], d['roofline_decode']['ms_per_decode_step'])"; grep -v amdgpu.ids $R/gpurun_out/bench.err | tail -3
