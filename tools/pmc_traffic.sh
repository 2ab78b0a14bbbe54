#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the kernels of one benchmark script (GPU box).
# usage: bash tools/pmc_traffic.sh <tag> <script.py> [args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/traffic_$TAG
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python3 $ROOT/"$@" > $O/$c.out 2> $O/$c.err) || { tail -5 $O/$c.err; exit 1; }
done
cd $ROOT
python3 - <<PY
import csv, glob
agg = {}
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"], r["Dispatch_Id"])
        agg[k] = agg.get(k, 0) + float(r["Counter_Value"])
per = {}
for (kn, c, _), v in agg.items():
    per.setdefault((kn, c), []).append(v)
for (kn, c), v in sorted(per.items()):
    print("%-62s %-11s %10.1f MB per launch (KB units x 1024; %d launches)" % (kn, c, sum(v) / len(v) * 1024 / 1e6, len(v)))
PY
