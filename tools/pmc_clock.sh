#!/bin/bash
# Effective clock of the calling kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel time (microarch guide, DVFS give-back).
set -e
OUT=gpurun_out/clk_$1; shift
mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD; cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/p -- python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 "$@" > $ROOT/$OUT/b.json 2> $ROOT/$OUT/p.err || { tail -5 $ROOT/$OUT/p.err; exit 1; }
cd $ROOT
python3 - <<PY
import csv,glob
g={};t={}
for f in glob.glob('$OUT/p/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bsc_call' in r['Kernel_Name']: g[r['Dispatch_Id']]=g.get(r['Dispatch_Id'],0)+float(r['Counter_Value'])
for f in glob.glob('$OUT/p/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bsc_call' in r['Kernel_Name']: t[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for d in sorted(g, key=int):
    if d in t: print('dispatch',d,'GUI_ACTIVE',g[d],'ns',t[d],'clock GHz', g[d]/8/t[d])
PY
