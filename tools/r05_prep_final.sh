#!/bin/bash
# Device pre-processing, the numbers of record (GPU box): kernel trace of tools/bench_prep.py, the copy kernel's SQ counters,
# HBM traffic, then the BAM -> BCF pipeline both ways (tools/bench_bam2bcf.py).  usage: bash tools/r05_prep_final.sh <tag>
set -e
TAG=${1:-r05_prepf}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $ROOT/tools/bench_prep.py > $O/prep.json 2> $O/err.txt) || { tail $O/err.txt; exit 1; }
python3 tools/kstats_timed.py $O/t 2 > $O/prep_kernels_timed.txt
cp $(ls $O/t/*/*_kernel_stats.csv | head -1) $O/prep_kernel_stats.csv
python3 tools/bench_prep.py > $O/prep_plain.json 2>> $O/err.txt
bash tools/pmc_kernel.sh ${TAG} bsc_prep_copy_kernel tools/bench_prep.py --steps 2 > $O/prep_copy_sq_counters.txt 2>&1 || { tail $O/prep_copy_sq_counters.txt; exit 1; }
bash tools/pmc_traffic.sh ${TAG} tools/bench_prep.py --steps 2 > $O/prep_traffic.txt 2>&1 || { tail $O/prep_traffic.txt; exit 1; }
python3 tools/bench_bam2bcf.py 2000000 > $O/bam2bcf.json 2> $O/bam2bcf.err || { tail $O/bam2bcf.err; exit 1; }
cat $O/prep_kernels_timed.txt $O/prep_plain.json $O/prep_copy_sq_counters.txt; grep bsc_ $O/prep_traffic.txt; cat $O/bam2bcf.json
