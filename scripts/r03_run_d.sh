#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03d; mkdir -p $OUT; cd $R
timeout -k 10 200 scripts/ubench/stream_shapes > $OUT/stream_shapes.txt 2>&1 || exit 1
echo shapes done
timeout -k 10 400 python -u -m pytest tests/test_gpu_north_star.py -x -q -s --timeout 300 --timeout-method thread > $OUT/north_star_tests.txt 2>&1; tail -15 $OUT/north_star_tests.txt
TOL=3.5147e-7
timeout -k 10 200 python bench.py --n 512 --tol $TOL --no-cpu-baseline > $OUT/bench512_matched_single.json 2> $OUT/bench512_matched_single.err || exit 1
timeout -k 10 200 python bench.py --n 512 --tol $TOL --no-cpu-baseline --local-slabs 8 > $OUT/bench512_matched_slabs8.json 2> $OUT/bench512_matched_slabs8.err || exit 1
FSAMD_NO_SWEEP_PAIRS=1 timeout -k 10 200 python bench.py --n 512 --tol $TOL --no-cpu-baseline --local-slabs 8 > $OUT/bench512_matched_slabs8_nopairs.json 2> $OUT/bench512_matched_slabs8_nopairs.err || exit 1
python - <<PY
import json
for f in ("single","slabs8","slabs8_nopairs"):
    l=json.load(open("$OUT/bench512_matched_%s.json"%f))
    print(f, l["ms_per_step"], l["config"]["sweeps_per_step"], l["config"]["residual_passes_per_step"])
PY
