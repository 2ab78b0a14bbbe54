#!/bin/bash
# A/B of library build variants on the grouping kernels (bsc_bin_count_kernel / bsc_bin_scatter_kernel) and the walk kernels:
# kernel trace of tools/bench_reads.py per variant.  usage (GPU box): bash tools/ab_bin.sh <tag> <variant names... | main>
set -e
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
for v in "$@"; do
  if [ $v != main ]; then export BSCALL_AMD_LIB=$ROOT/bs_call_amd/lib/variants/lib_$v.so; else unset BSCALL_AMD_LIB; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -- python3 $ROOT/tools/bench_reads.py --steps 6 --warm 6 --no-check ${BENCH_ARGS} > $O/$v.json 2> $O/$v.err) || { tail -5 $O/$v.err; exit 1; }
  echo "== $v"
  python3 $ROOT/tools/kstats_timed.py $O/trace_$v 6 bsc_ | grep -v '^#' | cut -c1-70,95-140 | tee -a $O/summary.txt
done
