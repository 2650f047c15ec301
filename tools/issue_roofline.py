"""Instruction-issue view of the three hot kernels from rocprofv3 --pmc passes of `bench.py --timed-only` (tools/measure_round.sh):
why the HBM roofline fraction of a kernel is what it is.

    python tools/issue_roofline.py <out.json> <counter_collection.csv> [...]
    python tools/issue_roofline.py <out.json> --from-json <earlier out.json>     (recompute the ratios from its per-launch sums)

Per kernel family (k_light_sweep, k_light_occlusion, k_raymarch_lit; averaged per launch, summed over the chip). GRBM_GUI_ACTIVE
is summed over the 8 XCDs (a 0.58 ms frame reads 11.0 M = 8 x 0.58 ms x 2.37 GHz), so a launch lasts GRBM_GUI_ACTIVE / 8 cycles;
SQ_WAVE_CYCLES and SQ_WAIT_INST_ANY count in units of 4 cycles.
  valu_issue_frac   SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch cycles): share of the chip's VALU issue slots used, at the
                    4 cycles per wave64 instruction the round-2 review prescribes (the round-2 figure for the frame, 0.86, is
                    this ratio)
  salu_per_valu     scalar per vector instruction (a wave issues one instruction at a time: scalars cost issue slots too)
  lds_busy_frac     SQ_ACTIVE_INST_LDS x 4 / (256 CUs x launch cycles) (approximate: the counter's unit is not documented)
  wait_share        SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of their resident time the waves spend waiting (memory, LDS, barrier)
  waves_per_simd    SQ_WAVE_CYCLES x 4 / (1024 x launch cycles): average resident waves
"""
import collections
import csv
import json
import sys

FAMILIES = ("k_light_sweep", "k_light_occlusion", "k_raymarch_lit", "k_light_chain")


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if len(sys.argv) > 3 and sys.argv[2] == "--from-json":
        for fam, e in json.load(open(sys.argv[3])).items():
            for k, v in e["per_launch"].items():
                acc[fam][k] = [e["launches"], v * e["launches"]]
        sys.argv[2:] = []
    for path in sys.argv[2:]:
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            fam = next((f for f in FAMILIES if f in name), None)
            if fam is None:
                continue
            a = acc[fam][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    out = {}
    for fam, counters in acc.items():
        c = {k: v[1] / max(v[0], 1) for k, v in counters.items()}
        gui = c.get("GRBM_GUI_ACTIVE")
        gui = gui / 8.0 if gui else gui  # cycles of one launch
        entry = {"launches": max(v[0] for v in counters.values()), "per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
        if gui:
            if "SQ_INSTS_VALU" in c:
                entry["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * gui), 4)
            if "SQ_INSTS_SALU" in c and c.get("SQ_INSTS_VALU"):
                entry["salu_per_valu"] = round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3)
            if "SQ_ACTIVE_INST_LDS" in c:
                entry["lds_busy_frac"] = round(c["SQ_ACTIVE_INST_LDS"] * 4.0 / (256.0 * gui), 4)
            if "SQ_WAVE_CYCLES" in c:
                entry["waves_per_simd"] = round(c["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * gui), 3)
            entry["launch_us_at_2_4_ghz"] = round(gui / 2400.0, 1)
        if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in c:
            entry["wait_share"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        out[fam] = entry
    csvs = [a for a in sys.argv[2:] if a.endswith(".csv")]
    if csvs:  # what it was collected from (bench.py quotes it only for the same kernel sources and launches per step)
        import os

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from pmc_traffic import signature

        first = next(csv.DictReader(open(csvs[0])))["Counter_Name"]
        out["_signature"] = signature(csvs[0], first)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    for fam, e in out.items():
        if fam.startswith("_"):
            print(fam, e)
            continue
        print(fam, {k: v for k, v in e.items() if k != "per_launch"})
        for k, v in e["per_launch"].items():
            print(f"    {k:28s} {v:18.1f}")


if __name__ == "__main__":
    main()
