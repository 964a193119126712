cd /tmp && export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5d; mkdir -p $O
DOTS_OCR_GEMM_PLAN=1 DOTS_OCR_LIB=$R/tools/bin/var_w4_prof/libdots_ocr_hip.so timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/w4_prof.txt
python - $O/w4_prof.txt <<'PY'
import re,sys,collections
d=collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m=re.search(r"epi (\d+) N (\d+) K (\d+) blk (\d+): prologue (\d+) cyc / ([\d.]+) us, main loop (\d+) cyc / ([\d.]+) us \((\d+) cyc per K tile\), epilogue \+ drain (\d+) cyc / ([\d.]+) us; issue (\d+) drain (\d+)",ln)
    if m: d[(m.group(1),m.group(2),m.group(3))].append([float(x) for x in m.groups()[4:]])
for k,v in d.items():
    n=len(v); avg=[sum(x[i] for x in v)/n for i in range(9)]
    print("epi %s N %s K %s (%d samples): prologue %.0f cyc %.2f us | main %.0f cyc %.2f us, %.0f cyc/Ktile | epilogue %.0f cyc %.2f us (issue %.0f, drain %.0f)"%(k+(n,)+tuple(avg)))
PY
