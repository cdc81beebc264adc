#!/bin/bash
# The pass of two sweeps on z-slabs with the planes next to both interfaces of a slab in ONE launch: the slab tests,
# then 512^3 at the tolerance matched to the 256^3 divergence as a single domain and as 8 and 2 local slabs.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03ends; mkdir -p $OUT; cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_sweep_pairs.py tests/test_gpu_slab_ranks.py -x -q --timeout 300 --timeout-method thread > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
TOL=3.5147e-7
LEGS="--no-cpu-baseline --no-multigrid-leg --no-512-leg --no-matched-leg --no-steady-leg"
for v in single slabs8 slabs2; do
  case $v in single) A="";; slabs8) A="--local-slabs 8";; slabs2) A="--local-slabs 2";; esac
  timeout -k 10 200 python bench.py --n 512 --tol $TOL $LEGS $A > $OUT/$v.json 2> $OUT/$v.err || { tail $OUT/$v.err; exit 1; }
done
python - <<PY
import json
for f in ("single","slabs8","slabs2"):
    l=json.load(open("$OUT/%s.json"%f))
    print(f, "%.2f ms/step" % l["ms_per_step"], l["config"]["sweeps_per_step"], l["config"]["residual_passes_per_step"])
PY
