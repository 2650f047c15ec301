#!/bin/bash
# PC sampling of the hot kernels (rocprofv3 --pc-sampling-beta-enabled; run on the GPU box through gpurun).
#   $1 = output name under gpurun_out/, $2.. = bench.py arguments (default: --raymarch-only)
# Tries the stochastic (hardware) method first and host_trap second; every attempt's stderr tail is kept so that a stack that
# cannot sample says so in the committed file. Output: gpurun_out/$1/{stochastic,host_trap}.{txt,log}: per kernel and instruction
# the sample counts (tools/pcsamp_summary.py).
OUT=gpurun_out/${1:-pcsamp}
shift
ARGS=${@:---raymarch-only}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for method in stochastic host_trap; do
  rm -rf /tmp/pcs_$method
  if [ $method = stochastic ]; then unit="--pc-sampling-unit cycles --pc-sampling-interval ${PCS_INTERVAL:-65536}"; else unit="--pc-sampling-unit time --pc-sampling-interval ${PCS_INTERVAL_US:-1}"; fi
  (cd /tmp && timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method $unit --kernel-trace --output-format csv json -d /tmp/pcs_$method -o p -- \
      python $ROOT/bench.py $ARGS --no-cpu-baseline --timed-only --steps ${PCS_STEPS:-40} --warmup 3 > "$ROOT/$OUT/$method.bench.json" 2> /tmp/pcs_$method.err)
  echo "rc=$?" > "$OUT/$method.log"
  grep -v amdgpu.ids /tmp/pcs_$method.err | tail -25 >> "$OUT/$method.log"
  find /tmp/pcs_$method -type f | sed 's/^/file: /' >> "$OUT/$method.log"
  python tools/pcsamp_summary.py /tmp/pcs_$method > "$OUT/$method.txt" 2>> "$OUT/$method.log"
  head -c 3000 "$OUT/$method.txt"
  tail -5 "$OUT/$method.log"
done
