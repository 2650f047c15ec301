#!/bin/bash
# PMC passes of the lit frame alone (bench.py --raymarch-only): where a frame's cycles go — VALU / VMEM / LDS / scalar issue activity,
# back-pressure from the texture-address path. (Derived TA_* / TCP_*_sum counter passes of this command did not finish within ten minutes
# on the round-5 box and are not collected.) Output: gpurun_out/$1/frame_pmc_<n>.csv (k_raymarch_lit rows only); round 5's summary: profiles/r05_frame_issue_detail.txt.
OUT=gpurun_out/${1:-frame_pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/fp_$i
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d /tmp/fp_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --raymarch-only --no-cpu-baseline --timed-only --steps 10 --warmup 2 > /dev/null 2>&1)
  f=$(find /tmp/fp_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep k_raymarch_lit "$f") > "$OUT/frame_pmc_$i.csv"; else echo "pass $i: no output ($set)"; fi
done
ls -la "$OUT"
