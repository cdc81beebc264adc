#!/bin/bash
# Measures red-black against lexicographic sweep order after 100 steps of the vortex ring at several sizes and stopping rules.
# Usage (on the GPU box): bash scripts/north_star_table.sh > gpurun_out/north_star_table.txt
set -o pipefail
run() { timeout -k 10 "$1" python -u scripts/north_star_orders.py "${@:2}" | grep --line-buffered RESULT; }
run 200 64 100 --rule l1:5 --check-every 1 || exit 1
run 200 64 100 --rule fixed:300 || exit 1
run 300 64 100 --rule fixed:1000 || exit 1
run 200 64 100 --rule fixed:300 --free 20 || exit 1
run 300 64 100 --rule fixed:1000 --free 20 || exit 1
run 400 128 100 --rule l1:5 --check-every 25 || exit 1
run 400 128 100 --rule fixed:1000 || exit 1
run 400 128 100 --rule fixed:1000 --free 20 || exit 1
