"""For every kernel of a rocprofv3 --kernel-trace --hip-runtime-trace run: when the host called hipLaunchKernel against when
the kernel started on the device (joined by correlation id) — was a late start the host's or the device's doing?
    python tools/launch_vs_start.py <hip_api_trace.csv> <kernel_trace.csv> [name filter]"""
import csv
import re
import sys

api = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "Launch" in r["Function"]:
            api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = []
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("tbrm::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Correlation_Id"], r.get("Queue_Id", "?")))
rows.sort()
flt = sys.argv[3] if len(sys.argv) > 3 else ""
t0 = rows[0][0]
prev_end = {}
last_end = None
print(f"{'start us':>10s} {'dur us':>8s} {'q':>2s} {'host call -> start us':>22s} {'prev kernel end -> start us':>28s}  kernel")
for s, e, n, cid, q in rows[-int(sys.argv[4]) if len(sys.argv) > 4 else 0:]:
    a = api.get(cid)
    if flt in n:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {q:>2s} {((s - a[0]) / 1e3 if a else float('nan')):22.1f} {((s - last_end) / 1e3 if last_end else 0.0):28.1f}  {n}")
    last_end = max(last_end or e, e)
