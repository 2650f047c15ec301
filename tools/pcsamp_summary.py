"""Summary of a rocprofv3 PC-sampling run (tools/pc_sampling.sh): per kernel the samples per instruction, and — stochastic
samples — how many of them found the wave issuing, the instruction types and the reasons for not issuing.

usage: pcsamp_summary.py <rocprofv3 output dir> [top instructions per kernel, default 70]"""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 70
    kernels = {}
    for f in find(root, "*kernel_trace.csv"):
        for row in csv.DictReader(open(f, newline="")):
            kernels[row.get("Dispatch_Id")] = short(row.get("Kernel_Name", "?"))
    files = [f for f in find(root, "*pc_sampling*.csv")]
    if not files:
        print("no pc_sampling csv under", root)
        return
    for f in files:
        rd = csv.DictReader(open(f, newline=""))
        cols = rd.fieldnames or []
        print(f"== {os.path.basename(f)}: columns {cols}")
        low = {c.lower(): c for c in cols}
        c_inst = low.get("instruction")
        c_cmt = low.get("instruction_comment")
        c_disp = low.get("dispatch_id")
        c_issued = next((low[c] for c in low if "issued" in c), None)
        c_type = next((low[c] for c in low if c in ("instruction_type", "inst_type")), None)
        c_stall = next((low[c] for c in low if "stall" in c or "not_issued" in c), None)
        c_exec = low.get("exec_mask")
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0, collections.Counter(), 0]))
        totals = collections.Counter()
        types = collections.defaultdict(collections.Counter)
        stalls = collections.defaultdict(collections.Counter)
        n = 0
        for row in rd:
            n += 1
            k = kernels.get(row.get(c_disp), "dispatch " + str(row.get(c_disp)))
            key = (row.get(c_inst, "?"), row.get(c_cmt, "") if c_cmt else "")
            e = per[k][key]
            e[0] += 1
            issued = c_issued and str(row.get(c_issued)).strip() in ("1", "True", "true")
            if issued:
                e[1] += 1
            if c_stall:
                e[2][row.get(c_stall)] += 1
                if not issued:
                    stalls[k][row.get(c_stall)] += 1
            if c_exec:
                try:
                    e[3] += bin(int(row.get(c_exec), 0)).count("1")
                except (TypeError, ValueError):
                    pass
            if c_type:
                types[k][row.get(c_type)] += 1
            totals[k] += 1
        print(f"{n} samples")
        for k, cnt in totals.most_common(6):
            print(f"\n#### {k}: {cnt} samples ({100.0 * cnt / max(n, 1):.1f} % of all)")
            if c_issued:
                iss = sum(e[1] for e in per[k].values())
                print(f"  wave issued an instruction in {iss} samples ({100.0 * iss / cnt:.1f} %)")
            if types[k]:
                print("  instruction types:", ", ".join(f"{t} {100.0 * c / cnt:.1f} %" for t, c in types[k].most_common(8)))
            if stalls[k]:
                tot = sum(stalls[k].values())
                print("  not issued because:", ", ".join(f"{t} {100.0 * c / max(tot, 1):.1f} %" for t, c in stalls[k].most_common(8)))
            print(f"  {'samples':>8} {'%':>6} {'issued%':>8} {'lanes':>6}  instruction  [source]  {{top stall reasons}}")
            for (inst, cmt), e in sorted(per[k].items(), key=lambda kv: -kv[1][0])[:top]:
                why = ", ".join(f"{t}:{c}" for t, c in e[2].most_common(3)) if c_stall else ""
                lanes = e[3] / e[0] if e[0] and c_exec else 0.0
                print(f"  {e[0]:8d} {100.0 * e[0] / cnt:6.2f} {100.0 * e[1] / e[0] if c_issued else 0.0:8.1f} {lanes:6.1f}  {inst}  [{os.path.basename(cmt) if cmt else ''}]  {{{why}}}")


if __name__ == "__main__":
    main()
