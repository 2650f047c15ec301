"""Condenses tools/sweep_stamps.py output: per stamped launch the free tile's pace, the lag per hop at slice 63 and at the last
slice, and when the last tile ended. Reads stdin."""
import re
import sys

title, rows = None, []


def flush():
    if rows:
        h0, hl = rows[0], rows[-1]
        n = hl[0] - h0[0]
        pace = (h0[3] - h0[2]) / 448.0
        print(f"{title or '':70s} hops {n:3d}  free tile {pace:.3f} us/slice  lag/hop: start {(hl[1] - h0[1]) / n:.2f} slice63 {(hl[2] - h0[2]) / n:.2f} "
              f"last {(hl[3] - h0[3]) / n:.2f} us  first tile done {h0[3]:.1f}  end {hl[4]:.1f} us")
    rows.clear()


for line in sys.stdin:
    if line.startswith("=="):
        flush()
        title = line.strip()
    elif line.startswith("[tbrm sweep stamps]"):
        flush()
        title = (title or "") + " | " + line.split("]")[1].split(",")[0].strip()
    else:
        m = re.match(r"\s*hop\s+(\d+):\s+\d+ tiles\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            rows.append((int(m.group(1)),) + tuple(float(m.group(i)) for i in range(2, 6)))
flush()
