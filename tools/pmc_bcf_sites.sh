#!/bin/bash
# SQ counter passes for the BCF encoder's per-position form (bsc_bcf_write_kernel).  usage: bash tools/pmc_bcf_sites.sh <tag>
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/bench_sites_bcf.py --steps 2 > $OUT/b1.json 2> $OUT/p1.err || { tail -5 $OUT/p1.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/bench_sites_bcf.py --steps 2 > $OUT/b2.json 2> $OUT/p2.err || { tail -5 $OUT/p2.err; exit 1; }
cd $ROOT
python3 - $OUT <<'PY' | tee $OUT/bcf_write_sq_counters.txt
import csv, glob, sys
out = sys.argv[1]
agg = {}
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bsc_bcf_write' not in r['Kernel_Name']: continue
        k = (r['Counter_Name'], r['Dispatch_Id'])
        agg[k] = agg.get(k, 0) + float(r['Counter_Value'])
per = {}
for (c, _), v in agg.items(): per.setdefault(c, []).append(v)
print('# bsc_bcf_write_kernel, per-position form, 50 M positions at 30x: SQ counters per launch (mean over the launches of the run)')
for c, v in sorted(per.items()): print('%-24s %.4g' % (c, sum(v) / len(v)))
PY
