"""Launch gaps from a rocprofv3 --kernel-trace CSV: for every pair of consecutive kernels on the same queue, the idle
time between the end of one and the start of the next, grouped by (previous kernel, next kernel).

    python tools/trace_gaps.py <..._kernel_trace.csv> [max_gap_us]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("tbrm::", "")


def main():
    path = sys.argv[1]
    max_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    gaps = defaultdict(list)
    busy = defaultdict(float)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = (s1 - e0) / 1e3
        if g < max_gap:  # larger: host-side pauses between bench steps
            gaps[(n0, n1)].append(g)
    for s, e, n in rows:
        busy[n] += (e - s) / 1e3
    print(f"{'previous -> next':90s} {'count':>6s} {'mean us':>8s} {'total ms':>9s}")
    tot = 0.0
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        tot += sum(v)
        print(f"{(k[0] + ' -> ' + k[1])[:90]:90s} {len(v):6d} {sum(v) / len(v):8.2f} {sum(v) / 1e3:9.3f}")
    print(f"total gap time {tot / 1e3:.3f} ms; total kernel time {sum(busy.values()) / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
