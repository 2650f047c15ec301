"""Time-ordered kernel list from a rocprofv3 --kernel-trace CSV: the last `count` kernels before the end (or those after
the last k_raymarch / between markers), with start offset, duration and the gap to the previous kernel.

    python tools/trace_timeline.py <kernel_trace.csv> [count]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("tbrm::", "")


def main():
    path = sys.argv[1]
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                         r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), r.get("VGPR_Count", "?"), r.get("Grid_Size", "?")))
    rows.sort()
    rows = rows[-count:]
    t0 = rows[0][0]
    prev_end = None
    print(f"{'start us':>10s} {'dur us':>8s} {'gap us':>8s} {'q':>3s} {'lds':>7s} {'vgpr':>5s} {'grid':>8s} kernel")
    for s, e, n, q, lds, vg, grid in rows:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t0) / 1e3:10.2f} {(e - s) / 1e3:8.2f} {gap:8.2f} {q:>3s} {lds:>7s} {vg:>5s} {grid:>8s} {n}")
        prev_end = max(prev_end or e, e)


if __name__ == "__main__":
    main()
