cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_kernels_gpu.py $R/tests/test_fullsize_gpu.py -x -q -k "attn or flash or causal" 2>&1 | tail -3
python $R/tools/microbench.py flash --iters 10
python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --max-new-tokens 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['phase_ms_per_step'], d['roofline']['achieved'], d['roofline_vit']['achieved'])"
