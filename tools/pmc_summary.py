#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, per counter mean over dispatches.
    python tools/pmc_summary.py gpurun_out/pmc/p1_counter_collection.csv [...]"""
import csv
import sys
from collections import defaultdict


def main():
    for path in sys.argv[1:]:
        acc = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("# " + path)
        for k, cs in acc.items():
            if "rocclr" in k:
                continue
            for c, v in sorted(cs.items()):
                print("%-62s %-24s n=%-4d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
