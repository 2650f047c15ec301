#!/bin/bash
# usage: ab.sh outdir "ENV1" "ENV2" ... ; each run: bench --no-cpu-baseline
out=$1; shift
mkdir -p gpurun_out/$out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$out/r$i.json
  python - "$e" gpurun_out/$out/r$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read()); g=d["gpu_ms"]
print(sys.argv[1], "step", d["ms_per_step"], "ray", g["raymarch"], "chg", g["change_dir_light"], "unc", g["change_dir_light_uncached"], "cold", g["reset_all_lights_cold"], "warm", g["reset_all_lights_warm"], "win", g["window_sweep_step"])
PY
done
