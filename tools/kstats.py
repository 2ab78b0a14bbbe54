#!/usr/bin/env python3
"""rocprofv3's kernel_stats.csv as a short table: tools/kstats.py FILE [n]"""
import csv
import sys

for k, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if k >= (int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        break
    print("%-44s calls %-4s avg %10.1f us  min %10.1f  max %10.1f" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
