#!/bin/bash
# SQ counter passes for the fused chain kernel (run on the GPU box via gpurun). usage: tools/pmc_chain.sh <tag> [bench_chain args]
set -e
TAG=$1; shift
OUT=gpurun_out/sqc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/$OUT/p1 -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2 "$@" > $ROOT/$OUT/b1.json 2> $ROOT/$OUT/p1.err || { tail -5 $ROOT/$OUT/p1.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $ROOT/$OUT/p2 -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2 "$@" > $ROOT/$OUT/b2.json 2> $ROOT/$OUT/p2.err || { tail -5 $ROOT/$OUT/p2.err; exit 1; }
cd $ROOT
python3 - <<PY
import csv,glob
agg={}
for f in glob.glob('$OUT/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bsc_chain_kernel_t<true' not in r['Kernel_Name'] and 'chain_kernel_tILb1' not in r['Kernel_Name']: continue
        k=(r['Counter_Name'],r['Dispatch_Id'])
        agg[k]=agg.get(k,0)+float(r['Counter_Value'])
per={}
for (c,_),v in agg.items(): per.setdefault(c,[]).append(v)
print("# bsc_chain_kernel_t<true>, 50 M positions at 30x with statistics, mean per launch")
for c,v in sorted(per.items()): print('%-24s %.4g'%(c,sum(v)/len(v)))
PY
