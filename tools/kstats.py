#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: usage tools/kstats.py <file-or-dir> [min_us]"""
import csv, glob, os, re, sys
p = sys.argv[1]
fs = [p] if os.path.isfile(p) else glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True)
for f in fs:
    print("#", f)
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        n = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "", n)
        n = re.sub(r"\(.*", "", n)[:90]
        print("%-92s calls %4s  avg %10.1f us  min %10.1f  total %8.2f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
