#!/bin/bash
# Mean duration of the occlusion kernels of uncached Adds at bench size (rocprofv3 kernel trace of tools/sweep_time.py, cache off),
# per library variant: tools/occ_time.sh "" tools/tmp/exp/libtbrm_occ91.so ...
export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/prof_occ
  ( [ -n "$lib" ] && export TBRM_LIB_PATH=$PWD/$lib; VARIANTS="light_cache_mb=0,light_sweep=${SWEEP:-1}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_occ -o p -- python tools/sweep_time.py > /dev/null 2>&1 )
  echo "== ${lib:-product}"
  f=$(find /tmp/prof_occ -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_light_occlusion" in n or "k_occ_" in n:
        print(f'  {n[:60]:60s} calls {r["Calls"]:>4s} mean {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
done
