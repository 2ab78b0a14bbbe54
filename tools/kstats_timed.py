#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 --kernel-trace directory, with the first `skip` launches of every kernel
(the warm-ups of the traced command) left out of the average — so that the tracked number and the HIP-event average of the
same command's timed region can be compared.  usage: tools/kstats_timed.py <dir> [skip] [name-substring]"""
import csv, glob, os, re, sys
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sub = sys.argv[3] if len(sys.argv) > 3 else ""
per = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sub and sub not in n:
            continue
        per.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("# %s: durations in us; `timed` leaves out each kernel's first %d launches" % (d, skip))
print("%-70s %6s %10s %10s %10s | %6s %10s %10s %10s" % ("kernel", "calls", "avg", "min", "max", "timed", "avg", "min", "max"))
for n, v in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    v.sort()
    a = [x[1] / 1e3 for x in v]
    t = a[skip:] if len(a) > skip else a
    short = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "", n)
    short = re.sub(r"\(.*", "", short)[:70]
    if sum(a) < 50 and "bsc_" not in n:
        continue
    print("%-70s %6d %10.1f %10.1f %10.1f | %6d %10.1f %10.1f %10.1f" % (short, len(a), sum(a) / len(a), min(a), max(a), len(t), sum(t) / len(t), min(t), max(t)))
