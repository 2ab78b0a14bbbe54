#!/bin/bash
# Round 5 (GPU box): the path the drop-in glue runs.  Parity tests of the block entries, the small-block bench (incl. the gt_vcf
# forms bsc_blocks_submit_to / _inplace), and the glue's own protocol end to end in integration/demo_block against mock printers
# of different speeds — round 5's protocol (lib/demo_block) beside round 4's (lib/variants/demo_block_r4, built from c576228).
# usage: bash tools/r05_glue.sh <tag>
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
[ -n "$SKIP_TESTS" ] || timeout -k 10 900 python3 -m pytest tests/test_gpu_blocks.py tests/test_gpu_accumulate.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -2 $O/pytest.txt
timeout -k 10 900 python3 tools/bench_small_blocks.py > $O/small_blocks.json 2> $O/small_blocks.txt || { tail -20 $O/small_blocks.txt; exit 1; }
cat $O/small_blocks.txt
: > $O/demo.txt
# BSC_DEMO_MPROF_JOBS=4: the mock profiling thread costs a mutex round trip per job; with one job per template it, not the glue, is
# what a run of small blocks waits for.  1 200 blocks of 10 000 positions: the page-locked arrays are allocated during the first batches.
for ns in -1 0 40; do
  for exe in bs_call_amd/lib/demo_block bs_call_amd/lib/variants/demo_block_r4; do
    [ -x $exe ] || continue
    for rep in 1 2; do
      echo "== $exe, mock printer $ns ns per position, ${DEMO_BLOCKS:-1200} blocks of 10 000 positions at 30x" >> $O/demo.txt
      BSC_DEMO_MPROF_JOBS=4 BSC_DEMO_PRINT_NS=$ns timeout -k 10 300 $exe 10000 30 ${DEMO_BLOCKS:-1200} 2>&1 | grep -E "blocks, |glue protocol end to end" >> $O/demo.txt
    done
  done
done
cat $O/demo.txt
