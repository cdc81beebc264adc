#!/bin/bash
# Collects, on the GPU box, what profiles/ holds: rocprofv3 kernel statistics of the default bench
# command at 256^3 and 512^3, and the FETCH_SIZE / WRITE_SIZE counter passes of the sweep kernel
# (one run per counter, never combined with tracing).  usage: collect_profiles.sh <tag>   (e.g. r01)
# Output goes to gpurun_out/profiles_<tag>/ ; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $OUT && mkdir -p $OUT   # (merged results of an earlier run with the same tag would be mixed in: use a fresh tag)
cd /tmp && export TMPDIR=/tmp
# the default command itself (256^3 headline + its multigrid and 512^3 legs; the CPU baseline adds no kernels)
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- \
    python3 $R/bench.py --no-cpu-baseline > $OUT/bench_default_line_under_rocprof.json
echo "stats default done"
for n in 256 512; do   # one size per run, so that the per-kernel averages are not mixed
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$n -- \
      python3 $R/bench.py --no-cpu-baseline --no-multigrid-leg --no-512-leg --no-steady-leg --n $n > $OUT/bench${n}_line_under_rocprof.json
  echo "stats $n done"
done
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mg256 -- \
    python3 $R/bench.py --no-cpu-baseline --no-512-leg --no-steady-leg --solver multigrid > $OUT/bench_mg256_line_under_rocprof.json
echo "stats multigrid done"
for n in 256 512; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/${c}_$n -- python3 $R/scripts/sweep_only.py $n
    echo "pmc $c $n done"
  done
done
