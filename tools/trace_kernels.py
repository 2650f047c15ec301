"""Per-kernel duration statistics from a rocprofv3 --kernel-trace CSV (tbrm kernels only).

    python tools/trace_kernels.py <..._kernel_trace.csv>
"""
import csv
import re
import sys
from collections import defaultdict


def main():
    d = defaultdict(list)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
            if "tbrm::" not in n:
                continue
            d[n.replace("tbrm::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"{'kernel':44s} {'n':>6s} {'mean us':>9s} {'median':>9s} {'min':>9s} {'max':>9s} {'total ms':>9s}")
    for n, v in sorted(d.items()):
        v = sorted(v)
        print(f"{n:44s} {len(v):6d} {sum(v) / len(v):9.2f} {v[len(v) // 2]:9.2f} {v[0]:9.2f} {v[-1]:9.2f} {sum(v) / 1e3:9.3f}")


if __name__ == "__main__":
    main()
