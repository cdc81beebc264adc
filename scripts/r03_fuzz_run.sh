#!/bin/bash
# The fuzzers after a change of the slab path: slabs against the single domain (three shape modes), ranks against the
# single domain, the library against the oracle.  usage: r03_fuzz_run.sh <outdir under gpurun_out> [seconds per fuzzer]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; T=${2:-90}
mkdir -p $OUT; cd $R
timeout -k 10 $((T+60)) python -u scripts/fuzz_slabs.py ${SEED:-101} $T > $OUT/fuzz_slabs.txt 2>&1 || { tail -5 $OUT/fuzz_slabs.txt; exit 1; }
tail -1 $OUT/fuzz_slabs.txt
timeout -k 10 $((T+60)) python -u scripts/fuzz_slabs.py $((${SEED:-101}+1)) $T wide > $OUT/fuzz_slabs_wide.txt 2>&1 || { tail -5 $OUT/fuzz_slabs_wide.txt; exit 1; }
tail -1 $OUT/fuzz_slabs_wide.txt
timeout -k 10 $((T+60)) python -u scripts/fuzz_slabs.py $((${SEED:-101}+2)) $T tall > $OUT/fuzz_slabs_tall.txt 2>&1 || { tail -5 $OUT/fuzz_slabs_tall.txt; exit 1; }
tail -1 $OUT/fuzz_slabs_tall.txt
timeout -k 10 $((T+120)) python -u scripts/fuzz_ranks.py $((${SEED:-101}+3)) $T > $OUT/fuzz_ranks.txt 2>&1 || { tail -5 $OUT/fuzz_ranks.txt; exit 1; }
tail -1 $OUT/fuzz_ranks.txt
timeout -k 10 $((T+60)) python -u scripts/fuzz_oracle.py $((${SEED:-101}+4)) $T > $OUT/fuzz_oracle.txt 2>&1 || { tail -5 $OUT/fuzz_oracle.txt; exit 1; }
tail -1 $OUT/fuzz_oracle.txt
