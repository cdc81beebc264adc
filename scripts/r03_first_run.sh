#!/bin/bash
# One GPU call: the GPU test suite, the fixed-point experiment, the copy yardstick, MacCormack kernel statistics.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
timeout -k 10 200 python -u scripts/fixed_point_orders.py 64 20 > $OUT/fixed_point_64.txt 2>&1 || { tail $OUT/fixed_point_64.txt; exit 1; }
timeout -k 10 200 python -u scripts/fixed_point_orders.py 64 20 --free > $OUT/fixed_point_64_free.txt 2>&1 || exit 1
timeout -k 10 200 python -u scripts/fixed_point_orders.py 32 20 > $OUT/fixed_point_32.txt 2>&1 || exit 1
echo fixed points done
timeout -k 10 120 scripts/ubench/copy_variants > $OUT/copy_variants.txt 2>&1 || exit 1
echo copies done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mac_stats -- python3 $R/scripts/time_maccormack.py 512 > $OUT/mac_line.txt 2>&1 || exit 1
python3 $R/scripts/kernel_stats.py $OUT/mac_stats 8
