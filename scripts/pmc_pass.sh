#!/bin/bash
# One rocprofv3 PMC pass over a short run of the sweep kernel.  usage: pmc_pass.sh <counter> <n> <outdir>
# (each counter gets a run of its own; never combined with tracing options)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $1 --output-format csv -d $3 -- python3 $R/scripts/sweep_only.py $2
