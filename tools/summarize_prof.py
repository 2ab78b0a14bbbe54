#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into the text summary kept under profiles/."""
import csv
import glob
import os
import sys

d = sys.argv[1]


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(d, pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


print("== rocprofv3 --kernel-trace --stats (kernel_stats) ==")
for r in rows("trace/**/*kernel_stats.csv"):
    print("%-28s calls=%-5s total_ns=%-14s avg_ns=%-12s min_ns=%-12s max_ns=%-12s pct=%s" % (
        r.get("Name", "")[:28], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MinNs"),
        r.get("MaxNs"), r.get("Percentage")))
print("== per-dispatch resources (kernel_trace) ==")
seen = set()
for r in rows("trace/**/*kernel_trace.csv"):
    k = r.get("Kernel_Name")
    if k in seen:
        continue
    seen.add(k)
    print("%-28s grid=%s wg=%s VGPR=%s accum=%s SGPR=%s LDS=%s scratch=%s" % (
        k[:28], r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
        r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    print("== PMC %s per dispatch (counter unit: KiB; FETCH_SIZE on gfx950 counts 1/2 of wide coalesced reads) ==" % name)
    agg = {}
    for r in rows(sub + "/**/*counter_collection.csv"):
        if r.get("Counter_Name") != name:
            continue
        key = (r.get("Kernel_Name"), r.get("Dispatch_Id"))
        agg[key] = agg.get(key, 0.0) + float(r.get("Counter_Value", 0))
    per_kernel = {}
    for (k, _), v in agg.items():
        per_kernel.setdefault(k, []).append(v)
    for k, vs in per_kernel.items():
        print("%-28s dispatches=%d avg_value=%.1f (KiB) => %.3f GB" % (k[:28], len(vs), sum(vs) / len(vs), sum(vs) / len(vs) * 1024 / 1e9))
