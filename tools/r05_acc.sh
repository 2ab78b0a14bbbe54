#!/bin/bash
# Round 5 (GPU box): the accumulate stage with tiles of 128 positions (default) against tiles of 64 (BSC_ACC_TILE64=1), same library:
# parity tests in both settings, then the kernels of tools/bench_reads.py under the kernel trace, alternating.
# usage: bash tools/r05_acc.sh <tag>
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T="tests/test_gpu_accumulate.py tests/test_gpu_reads_chain.py tests/test_gpu_fullsize.py tests/test_gpu_records.py"
[ -n "$SKIP_TESTS" ] || timeout -k 10 900 python3 -m pytest $T -x -q > $O/pytest128.txt 2>&1 || { tail -40 $O/pytest128.txt; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -2 $O/pytest128.txt
[ -n "$SKIP_TESTS" ] || BSC_ACC_TILE64=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_accumulate.py tests/test_gpu_reads_chain.py -x -q > $O/pytest64.txt 2>&1 || { tail -40 $O/pytest64.txt; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -2 $O/pytest64.txt
for rep in 1 2; do
for v in 128 64; do
  if [ $v = 64 ]; then export BSC_ACC_TILE64=1; else unset BSC_ACC_TILE64; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v.$rep -- python3 $ROOT/tools/bench_reads.py --steps 6 --warm 6 --no-check > $O/$v.$rep.json 2> $O/$v.$rep.err) || { tail -5 $O/$v.$rep.err; exit 1; }
  echo "== tiles of $v positions (pass $rep)"
  python3 $ROOT/tools/kstats_timed.py $O/trace_$v.$rep 6 bsc_ | grep -v '^#' | grep -E "accumulate|bin_" | cut -c1-70,95-140
  python3 -c "
import json
r = json.loads(open('$O/$v.$rep.json').read().strip().splitlines()[-1])
print('   stage: accumulate %.3f ms   reads -> records %.3f ms' % (r['accumulate']['device_ms_avg'], r['reads_chain']['device_ms_avg']))"
done
done
