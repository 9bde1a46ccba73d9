"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (gatmi kernels only).
usage: python tools/pmc_summary.py <dir-or-csv> [...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"gatmi::(\w+)", name)
    return m.group(1) if m else None


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for arg in sys.argv[1:]:
        files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if k:
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        parts = [f"{c}: avg {sum(v)/len(v):.1f} (n={len(v)})" for c, v in sorted(acc[k].items())]
        print(f"{k:28s} " + "  ".join(parts))


if __name__ == "__main__":
    main()
