#!/bin/bash
# Counter passes over a short driver script, one rocprofv3 run per group (never combined with tracing).
# usage: pmc_kernel.sh <outdir under gpurun_out> <driver.py> <arg> ; prints per-kernel averages.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; DRV=$R/scripts/$2; ARG=$3
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $P --output-format csv -d $OUT/g$i -- python3 $DRV $ARG > /dev/null 2>&1 || { echo "pass $i failed"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(sys.argv[1] + '/g*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(d)):
        acc[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    if 'fsamd' not in k: continue
    print(k)
    w = sum(d['SQ_WAVES']) / len(d['SQ_WAVES']) if 'SQ_WAVES' in d else 1
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"   {c:40s} {m:16.4g}   per wave {m / w:10.2f}")
PY
