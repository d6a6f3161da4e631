"""Aggregate rocprofv3 --pmc CSV output (one row per dispatch and counter) into per-kernel averages."""
import csv
import glob
import sys
from collections import defaultdict


def main(root):
    files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"][:60]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctrs in sorted(acc.items()):
        parts = [f"{c}: n={len(v)} avg={sum(v) / len(v):.4g}" for c, v in sorted(ctrs.items())]
        print(f"{name:62s} " + " | ".join(parts))


if __name__ == "__main__":
    main(sys.argv[1])
