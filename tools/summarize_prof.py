#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) per kernel: count, avg/total duration, counter means.
usage: tools/summarize_prof.py gpurun_out/prof_<tag>  > profiles/<name>.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for p in ("void alga::", "alga::"):
        if name.startswith(p):
            name = name[len(p):]
    return name[:60]


def main(root):
    print("# rocprofv3 summary of", root)
    for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        print("\n## kernel stats (%s)" % os.path.relpath(f, root))
        rows = list(csv.DictReader(open(f)))
        print("%-62s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows:
            print("%-62s %8s %14s %12.0f %8s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for f in files:
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("\n## counters (%s): mean per dispatch" % os.path.basename(d))
        for k in sorted(acc):
            parts = ["%s=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(acc[k].items())]
            print("%-40s %s" % (k, "  ".join(parts)))


if __name__ == "__main__":
    main(sys.argv[1])
