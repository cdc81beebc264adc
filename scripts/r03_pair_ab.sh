#!/bin/bash
# A/B of the two-sweep kernel on one box: the library in the tree against a saved build (exp/lib/libfs_base.so),
# after the pair tests have passed.  Timings of the two builds alternate so that drift of the box shows.
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_sweep_pairs.py -x -q --timeout 300 --timeout-method thread > gpurun_out/ab_tests.txt 2>&1 || { tail -30 gpurun_out/ab_tests.txt; exit 1; }
tail -2 gpurun_out/ab_tests.txt
for i in 1 2; do
  echo "== tree" && timeout -k 10 120 python -u scripts/time_pairs.py 256 512
  echo "== base" && FSAMD_LIB=$PWD/exp/lib/libfs_base.so timeout -k 10 120 python -u scripts/time_pairs.py 256 512
done
