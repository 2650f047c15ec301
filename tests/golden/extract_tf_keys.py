"""Recovers the colour-curve keys of the reference's shipped transfer functions (data, not code).

Run in the authoring container only (reads /root/reference/Content/Curves/TF_CT-*.uasset); writes
tests/golden/tf_curves.json, which is committed. A UCurveLinearColor holds four FRichCurves (R,G,B,A); each key is
serialised as a 27-byte record: InterpMode, TangentMode, TangentWeightMode (3 x uint8) + Time, Value,
ArriveTangent, ArriveTangentWeight, LeaveTangent, LeaveTangentWeight (6 x float32). Every shipped key has InterpMode 0 (RCIM_Linear) — SURVEY.md Appendix B.
"""
import glob
import json
import os
import struct
import sys

REC = 27


def parse_record(buf, off):
    if off + REC > len(buf):
        return None
    im, tm, wm = buf[off], buf[off + 1], buf[off + 2]
    if im > 3 or tm > 4 or wm > 3:
        return None
    t, v, at, aw, lt, lw = struct.unpack_from("<6f", buf, off + 3)
    for x in (t, v, at, aw, lt, lw):
        if x != x or abs(x) > 1e6:
            return None
    if not (-1e-6 <= t <= 1.0 + 1e-6) or not (-4.0 <= v <= 4.0):
        return None
    return im, t, v


def find_key_arrays(buf):
    """Finds runs of >= 2 plausible records whose times start at exactly 0 and increase strictly (the element
    count lives in the tagged-property header, not next to the data, so runs are found by content)."""
    arrays = []
    off = 0
    while off + 2 * REC <= len(buf):
        first = parse_record(buf, off)
        if first is not None and first[1] == 0.0:
            recs = [first]
            while True:
                nxt = parse_record(buf, off + len(recs) * REC)
                if nxt is None or not (nxt[1] > recs[-1][1]) or nxt[1] < 1e-4:
                    break
                recs.append(nxt)
            if len(recs) >= 2 and abs(recs[-1][1] - 1.0) < 1e-6:  # every shipped curve spans [0,1]
                arrays.append((off, recs))
                off += len(recs) * REC
                continue
        off += 1
    return arrays


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Content/Curves"
    out = {}
    for path in sorted(glob.glob(os.path.join(src, "TF_CT-*.uasset"))):
        buf = open(path, "rb").read()
        arrays = find_key_arrays(buf)
        if len(arrays) != 4:
            print(f"skip {os.path.basename(path)}: found {len(arrays)} key arrays", file=sys.stderr)
            continue
        name = os.path.basename(path)[:-len(".uasset")]
        out[name] = {
            "offsets": [a[0] for a in arrays],
            "interp_modes": sorted({r[0] for a in arrays for r in a[1]}),
            "channels": [{"times": [r[1] for r in a[1]], "values": [r[2] for r in a[1]]} for a in arrays],
        }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_curves.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {len(out)} curves to {dst}")


if __name__ == "__main__":
    main()
