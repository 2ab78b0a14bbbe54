#!/bin/bash
# kernel trace + per-kernel statistics of one command: tools/ktrace.sh OUT_DIR script.py [args]  (OUT_DIR under gpurun_out/; prints the kernel summary)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python3 $ROOT/"$@" > $ROOT/$OUT/out.txt 2> $ROOT/$OUT/err.txt) || { tail -5 $ROOT/$OUT/err.txt; exit 1; }
f=$(find $ROOT/$OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $ROOT/$OUT/kernel_stats.csv
cut -c1-220 $f | head -${KTRACE_LINES:-24}
