"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes) into per-kernel HBM traffic.

Usage: python tools/pmc_traffic.py <pmc_fetch.csv> <pmc_write.csv> <out.json>
Units and correction as MI355X_MICROARCH.md §HBM prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced stream, so reads are doubled. Reported per launch (average over launches).
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[name][0] += 1
        acc[name][1] += float(row["Counter_Value"])
    return acc


LIGHT_KERNELS = ("k_light_sweep", "k_light_chain", "k_light_occlusion", "k_occ_flags", "k_occ_compact", "k_propagate_slice")


def operator_runs(path, counter, scale):
    """Bytes of every ChangeDirLight call of the profiled command (bench.py --timed-only: setup, then per step one
    ChangeDirLight and one frame): in dispatch order, every maximal run of light-operator kernels AFTER the first frame is one
    call (the runs before it are the setup's ResetAllLights). Returns [(bytes, kind)], kind = which occlusion launches ran:
    'cached' (mode 3: only the new light's occlusion computed, the removed light's factors kept), 'both' (mode 1),
    'remove+add' (mode 0: the Add shader's rules), 'from cache' (no occlusion launch at all)."""
    rows = []
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0].replace("void ", "").strip(), float(row["Counter_Value"])))
    rows.sort()
    runs, cur, modes, seen_frame = [], 0.0, set(), False
    for _, name, value in rows:
        if "k_raymarch_lit" in name:
            if seen_frame and modes:
                kind = "cached" if "3" in modes else ("both" if "1" in modes else ("remove+add" if "0" in modes else "from cache"))
                runs.append((cur * scale, kind))
            seen_frame, cur, modes = True, 0.0, set()
        elif seen_frame and any(k in name for k in LIGHT_KERNELS):
            cur += value
            m = re.search(r"k_light_occlusion<\d+, (\d+),", name)
            if m:
                modes.add(m.group(1))
            elif not modes:
                modes.add("none")
    return runs


def signature(path, counter):
    """What the summary was collected from, so that bench.py can refuse to quote it for other kernels: the kernel sources' hash
    (bench.kernel_source_hash) and the launches per step of the hot kernel families AFTER the first frame (= the steps)."""
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash

    rows = []
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"]))
    rows.sort()
    counts, frames = collections.Counter(), 0
    for _, name in rows:
        if "k_raymarch_lit" in name:
            frames += 1
        if frames == 0:
            continue  # (the setup's ResetAllLights)
        for fam in ("k_light_sweep", "k_light_occlusion", "k_raymarch_lit"):
            if fam in name:
                counts[fam] += 1
    # (the light kernels behind the last frame belong to no step: none in bench.py --timed-only)
    steps = max(frames, 1)
    return {"kernel_source_hash": kernel_source_hash(), "steps": frames, "launches_per_step": {k: round(v / steps, 3) for k, v in sorted(counts.items())}}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        if "tbrm::" not in name:
            continue
        nf, f = fetch.get(name, [0, 0.0])
        nw, w = write.get(name, [0, 0.0])
        rd = 2.0 * f * 1024 / max(nf, 1)
        wr = w * 1024 / max(nw, 1)
        out[name] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                     "hbm_bytes_per_launch": rd + wr, "fetch_size_raw_kib": f / max(nf, 1), "write_size_raw_kib": w / max(nw, 1)}
    # per operator call (see operator_runs); reads doubled like the per-kernel figures
    reads, writes = operator_runs(sys.argv[1], "FETCH_SIZE", 2.0 * 1024), operator_runs(sys.argv[2], "WRITE_SIZE", 1024.0)
    ray = [v for k, v in out.items() if "k_raymarch_lit" in k]
    if ray and reads and len(reads) == len(writes):
        calls = [(r[0] + w[0], r[1]) for r, w in zip(reads, writes)]
        by_kind = collections.defaultdict(list)
        for b, kind in calls:
            by_kind[kind].append(b)
        out["_per_operator_call"] = {
            "change_dir_light_calls": len(calls),
            "change_dir_light_hbm_bytes": sum(b for b, _ in calls) / len(calls),
            "change_dir_light_hbm_bytes_by_kind": {k: {"calls": len(v), "hbm_bytes": sum(v) / len(v)} for k, v in sorted(by_kind.items())},
            "raymarch_hbm_bytes": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ray) / sum(v["launches"] for v in ray)}
    out["_signature"] = signature(sys.argv[1], "FETCH_SIZE")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if k.startswith("_"):
            print(k, v)
            continue
        print(f"{k:60s} x{v['launches']:5d}  read {v['read_bytes_per_launch']/1e6:9.2f} MB  write {v['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
