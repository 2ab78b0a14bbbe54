#!/bin/bash
# kernel-trace profile of the reads -> calls block path (tools/bench_block.py); prints the bsc_* kernel rows.
# usage (on the GPU box): bash tools/prof_block.sh [positions] [coverage]
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof_block; mkdir -p gpurun_out/prof_block
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_block -- python3 $R/tools/bench_block.py ${1:-8000000} ${2:-30} > $R/gpurun_out/bench_block.txt 2>&1 || exit 1
cd $R
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/prof_block/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "bsc" in r["Name"]: print(r["Name"][:30], r["Calls"], r["AverageNs"], r["Percentage"])
PY
