#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter values per kernel name over all *_counter_collection.csv files under a directory.

    python tools/pmc_by_kernel.py gpurun_out/pmc_dir
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void rtamd::", "")
                tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
                key = (name, row.get("Dispatch_Id"))
                if key not in seen:
                    seen.add(key)
                    n[name] += 1
    counters = sorted({c for k in tot.values() for c in k})
    print("kernel".ljust(28), "launches".rjust(9), *(c.rjust(20) for c in counters))
    for name in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        print(name[:28].ljust(28), str(n[name]).rjust(9), *(f"{tot[name].get(c, 0):20.4g}" for c in counters))


if __name__ == "__main__":
    main()
