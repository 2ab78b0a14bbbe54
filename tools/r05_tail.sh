#!/bin/bash
# Round 5 (GPU box): kernel times of the passes behind the chain kernel inside bsc_block_records / bsc_block_bcf, with and without the
# chain's emit bytes (BSC_NO_EMIT_BYTES), two rounds on one box.  usage: bash tools/r05_tail.sh <tag>
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
for rep in 1 2; do
for mode in on off; do
  if [ $mode = off ]; then export BSC_NO_EMIT_BYTES=1; else unset BSC_NO_EMIT_BYTES; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${mode}_$rep -- python3 $ROOT/tools/bench_tail.py > $O/run_${mode}_$rep.txt 2> $O/run_${mode}_$rep.err) || { tail -5 $O/run_${mode}_$rep.err; exit 1; }
  echo "== emit bytes $mode (round $rep): $(tail -1 $O/run_${mode}_$rep.txt)"
  python3 tools/kstats.py $O/prof_${mode}_$rep | grep -E "tile_emit|compact|bcf_|chain_kernel_t<true, false, false, true>|accumulate_kernel_t<true>" | cut -c1-60,93-
done
done
