#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (values in KB as rocprofv3 reports them).

usage: python tools/pmc_summary.py FETCH_SIZE=<dir> WRITE_SIZE=<dir> [--prefix aq_]
Prints, per counter, `kernel, dispatches, mean_KB, min_KB, max_KB` for our kernels, in the format kept
under profiles/ (MI355X_MICROARCH.md, HBM section: separate --pmc passes; FETCH_SIZE x2 for coalesced reads).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def summarise(counter, d, prefix):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                if prefix not in name:
                    continue
                acc[name[:60]].append(float(row["Counter_Value"]))
    print(f"[{counter}]  kernel, dispatches, mean_KB, min_KB, max_KB")
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name}, {len(v)}, {sum(v) / len(v):.1f}, {min(v):.1f}, {max(v):.1f}")
    print()


def main():
    prefix = "aq_"
    args = [a for a in sys.argv[1:]]
    if "--prefix" in args:
        i = args.index("--prefix")
        prefix = args[i + 1]
        del args[i:i + 2]
    for a in args:
        counter, d = a.split("=", 1)
        summarise(counter, d, prefix)


if __name__ == "__main__":
    main()
