#!/bin/bash
# Round-3 baseline on the GPU box: HOT LOOP A at config sizes (kernel-trace stats + SQ counters), 200x counters for the
# calling and chain kernels.  usage (gpurun): bash tools/prof_r03_baseline.sh <tag>
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
mkdir -p gpurun_out/$TAG
O=$ROOT/gpurun_out/$TAG
for CFG in "50000000 30" "10000000 200"; do
  set -- $CFG
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/acc_$2x -- python3 $ROOT/tools/bench_reads.py --sites $1 --coverage $2 --steps 10 > $O/acc_$2x.json 2> $O/acc_$2x.err || { tail -5 $O/acc_$2x.err; exit 1; }
  cd $ROOT
  cat $O/acc_$2x.json
  find $O/acc_$2x -name "*kernel_stats.csv" | xargs cat | cut -c1-220 > $O/acc_$2x_kernel_stats.csv
  cat $O/acc_$2x_kernel_stats.csv
  bash tools/pmc_kernel.sh ${TAG}_acc$2 bsc_accumulate_kernel tools/bench_reads.py --sites $1 --coverage $2 --steps 2 --no-check > $O/sq_acc_$2x.txt 2>&1 || { tail -20 $O/sq_acc_$2x.txt; exit 1; }
  cat $O/sq_acc_$2x.txt
done
bash tools/pmc_sq.sh ${TAG}_cfg4 --sites 10000000 --coverage 200 > $O/sq_call_200x.txt 2>&1 || { tail -20 $O/sq_call_200x.txt; exit 1; }
cat $O/sq_call_200x.txt
bash tools/pmc_chain.sh ${TAG}_cfg4 --sites 10000000 --coverage 200 > $O/sq_chain_200x.txt 2>&1 || { tail -20 $O/sq_chain_200x.txt; exit 1; }
cat $O/sq_chain_200x.txt
