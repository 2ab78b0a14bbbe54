#!/bin/bash
# SQ counter passes for one kernel of any of the tools/ benchmarks (run on the GPU box via gpurun).
# usage: tools/pmc_kernel.sh <tag> <kernel-name-substring> <script.py> [args...]
set -e
TAG=$1; KERN=$2; shift 2
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/$OUT/p1 -- python3 $ROOT/"$@" > $ROOT/$OUT/b1.txt 2> $ROOT/$OUT/p1.err || { tail -5 $ROOT/$OUT/p1.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $ROOT/$OUT/p2 -- python3 $ROOT/"$@" > $ROOT/$OUT/b2.txt 2> $ROOT/$OUT/p2.err || { tail -5 $ROOT/$OUT/p2.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM --output-format csv -d $ROOT/$OUT/p3 -- python3 $ROOT/"$@" > $ROOT/$OUT/b3.txt 2> $ROOT/$OUT/p3.err || { tail -5 $ROOT/$OUT/p3.err; exit 1; }
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
for c,v in sorted(per.items()): print('%-24s %.4g'%(c,sum(v)/len(v)))
PY
