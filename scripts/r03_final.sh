#!/bin/bash
# Closing run: the GPU suite, the bench lines of the default and the longer window, the kernel statistics under rocprofv3.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 400 python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err || exit 1
timeout -k 10 200 python bench.py --warmup 5 --steps 20 --no-multigrid-leg --no-512-leg --no-steady-leg > $OUT/bench_w5k20_line.json 2>/dev/null || exit 1
echo benches done
bash scripts/collect_profiles.sh $1_prof > $OUT/collect.log 2>&1 || { tail $OUT/collect.log; exit 1; }
echo profiles done
