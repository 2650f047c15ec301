#!/bin/bash
# L2 hit rate of the lit march with its pixel blocks dealt to the XCDs in launch order (ray_xcd_rows=0) and row by row (1):
# bash tools/l2_hits.sh <out dir>   (on the GPU box; rocprofv3 counter passes of bench.py --raymarch-only)
OUT=${1:-gpurun_out/l2}
mkdir -p "$OUT"; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/prof_l2_$v
  TBRM_RAY_XCD_ROWS=$v rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/prof_l2_$v -o p -- python bench.py --raymarch-only --no-cpu-baseline --timed-only --steps 10 --warmup 2 > /dev/null 2>&1
  find /tmp/prof_l2_$v -name "*counter_collection.csv" -exec cp {} "$OUT/l2_rows$v.csv" \;
done
python - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
for v in (0, 1):
    acc = collections.defaultdict(float); n = 0
    try:
        for row in csv.DictReader(open(f"{out}/l2_rows{v}.csv")):
            if "k_raymarch_lit" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
                n += row["Counter_Name"] == "TCC_HIT_sum"
    except Exception as e:
        print("rows", v, "unreadable:", e); continue
    h, m = acc.get("TCC_HIT_sum", 0), acc.get("TCC_MISS_sum", 0)
    print(f"ray_xcd_rows={v}: {n} frames; per frame L2 hits {h / max(n, 1):.3e} misses {m / max(n, 1):.3e} hit rate {h / max(h + m, 1):.4f}; "
          f"EA read requests {acc.get('TCC_EA0_RDREQ_sum', 0) / max(n, 1):.3e}")
PY
