#!/usr/bin/env python3
"""Condense the CSVs tools/profile/profile.sh collected into per-kernel numbers (avg duration, PMC sums per launch)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path


def main(d):
    d = Path(d)
    for f in sorted(d.glob("kt_*kernel_stats.csv")):
        print(f"== {f.name}")
        for row in csv.DictReader(open(f)):
            print("  {Name:60.60s} calls={Calls} avg_ns={AverageNs} min={MinNs} max={MaxNs} pct={Percentage}".format(**row))
    for f in sorted(d.glob("pmc_*counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:50]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
        print(f"== {f.name}")
        for k in acc:
            if "jpezy" not in k:
                continue
            for c in acc[k]:
                print(f"  {k:50s} {c:24s} per-launch={acc[k][c] / cnt[k][c]:.6g} launches={cnt[k][c]}")


if __name__ == "__main__":
    main(sys.argv[1])
