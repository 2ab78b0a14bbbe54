#!/bin/bash
# L2 / L1 cache counters for one kernel of any of the tools/ benchmarks (run on the GPU box via gpurun).
# usage: tools/pmc_cache.sh <tag> <kernel-name-substring> <script.py> [args...]
set -e
TAG=$1; KERN=$2; shift 2
OUT=gpurun_out/cache_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $ROOT/$OUT/p1 -- python3 $ROOT/"$@" > $ROOT/$OUT/b1.txt 2> $ROOT/$OUT/p1.err || { tail -5 $ROOT/$OUT/p1.err; exit 1; }
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $ROOT/$OUT/p2 -- python3 $ROOT/"$@" > $ROOT/$OUT/b2.txt 2> $ROOT/$OUT/p2.err || { tail -5 $ROOT/$OUT/p2.err; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/p3 -- python3 $ROOT/"$@" > $ROOT/$OUT/b3.txt 2> $ROOT/$OUT/p3.err || { tail -5 $ROOT/$OUT/p3.err; exit 1; }
cd $ROOT
python3 - <<PY
import csv,glob
agg={}
for f in glob.glob('$OUT/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if '$KERN' not in r['Kernel_Name']: continue
        k=(r['Counter_Name'],r['Dispatch_Id'])
        agg[k]=agg.get(k,0)+float(r['Counter_Value'])
per={}
for (c,_),v in agg.items(): per.setdefault(c,[]).append(v)
for c,v in sorted(per.items()): print('%-30s %.4g'%(c,sum(v)/len(v)))
PY
