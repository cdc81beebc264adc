#!/bin/bash
# Collects, on the GPU box, what profiles/ holds: rocprofv3 kernel statistics of the default bench
# command at 256^3 and 512^3, and the FETCH_SIZE / WRITE_SIZE counter passes of the sweep kernel
# (one run per counter, never combined with tracing).  usage: collect_profiles.sh <tag>   (e.g. r01)
# Output goes to gpurun_out/profiles_<tag>/ ; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 256 512; do
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$n -- \
      python3 $R/bench.py --no-cpu-baseline --n $n > $OUT/bench${n}_line_under_rocprof.json
  echo "stats $n done"
done
for n in 256 512; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/${c}_$n -- python3 $R/scripts/sweep_only.py $n
    echo "pmc $c $n done"
  done
done
