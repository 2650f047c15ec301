"""Per-kernel sums of every counter in rocprofv3 --pmc CSVs (tbrm kernels only), averaged per launch.

    python tools/pmc_summary.py <counter_collection.csv> [...]
"""
import collections
import csv
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in sys.argv[1:]:
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if "tbrm::" not in name:
                continue
            a = acc[name.replace("tbrm::", "")][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    for k in sorted(acc):
        print(k)
        for c, (n, v) in sorted(acc[k].items()):
            print(f"    {c:32s} launches {n:5d}  per launch {v / n:16.1f}")


if __name__ == "__main__":
    main()
