#!/bin/bash
# usage: tools/prof_change.sh <outdir> [ENV=VAL ...]   kernel trace (last Change as a timeline) + one SQ counter pass
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
TAG=$(echo "$@" | tr ' =' '__')
rm -rf /tmp/kt /tmp/pmc
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python tools/one_change.py > "$OUT/run_$TAG.txt" 2>&1
find /tmp/kt -name "*kernel_trace.csv" -exec cp {} /tmp/kt.csv \;
python tools/trace_kernels.py /tmp/kt.csv > "$OUT/kernels_$TAG.txt" 2>&1
python tools/trace_timeline.py /tmp/kt.csv 150 > "$OUT/timeline_$TAG.txt" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc -o p -- python tools/one_change.py > /dev/null 2>&1
find /tmp/pmc -name "*counter_collection.csv" -exec cp {} /tmp/pmc.csv \;
python tools/pmc_summary.py /tmp/pmc.csv > "$OUT/pmc_$TAG.txt" 2>&1
grep "change L" "$OUT/run_$TAG.txt"
