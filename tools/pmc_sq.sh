#!/bin/bash
# SQ counter pass for the calling kernel (run on the GPU box via gpurun). usage: tools/pmc_sq.sh <tag> [--sites N]
set -e
TAG=$1; shift
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/$OUT/p1 -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --steps 2 --warmup 1 "$@" > $ROOT/$OUT/b1.json 2> $ROOT/$OUT/p1.err || { tail -5 $ROOT/$OUT/p1.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $ROOT/$OUT/p2 -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --steps 2 --warmup 1 "$@" > $ROOT/$OUT/b2.json 2> $ROOT/$OUT/p2.err || { tail -5 $ROOT/$OUT/p2.err; exit 1; }
cd $ROOT
python3 - <<PY
import csv,glob
agg={}
for f in glob.glob('$OUT/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bsc_call' not in r['Kernel_Name']: continue
        k=(r['Counter_Name'],r['Dispatch_Id'])
        agg[k]=agg.get(k,0)+float(r['Counter_Value'])
per={}
for (c,_),v in agg.items(): per.setdefault(c,[]).append(v)
for c,v in sorted(per.items()): print('%-24s %.4g'%(c,sum(v)/len(v)))
PY
