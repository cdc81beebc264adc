#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e; mkdir -p $OUT; cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_north_star.py -x -q -s --timeout 300 --timeout-method thread > $OUT/north_star_tests.txt 2>&1; tail -12 $OUT/north_star_tests.txt
TOL=3.5147e-7
timeout -k 10 200 python bench.py --n 512 --tol $TOL --no-cpu-baseline --local-slabs 2 > $OUT/bench512_matched_slabs2.json 2> $OUT/bench512_matched_slabs2.err || exit 1
python - <<PY
import json
l=json.load(open("$OUT/bench512_matched_slabs2.json"))
print("slabs2", l["ms_per_step"], l["config"]["sweeps_per_step"], l["roofline"]["launch_us"] if "roofline" in l else None, l["roofline"].get("velocity_copy_us"))
PY
