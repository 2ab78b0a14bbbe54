#!/bin/bash
# Round 5 working script (GPU box): parity tests of the calling / chain kernels, then A/B timings of library variants
# (VARIANTS="main head ...") on the pile-up-in chain, the reads path and the calling kernel, alternating, one box.
# usage: bash tools/r05_step.sh <tag>     (SKIP_TESTS=1: timings only; TESTS="..." overrides the test list; REPS=2)
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T=${TESTS:-"tests/test_gpu_parity.py tests/test_gpu_accumulate.py tests/test_gpu_chain.py tests/test_gpu_reads_chain.py tests/test_gpu_records.py tests/test_gpu_fullsize.py tests/test_gpu_blocks.py"}
[ -n "$SKIP_TESTS" ] || timeout -k 10 1000 python3 -m pytest $T -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -3 $O/pytest.txt
for rep in $(seq 1 ${REPS:-2}); do
for v in ${VARIANTS:-main}; do
  if [ $v != main ]; then export BSCALL_AMD_LIB=$ROOT/bs_call_amd/lib/variants/lib_$v.so; else unset BSCALL_AMD_LIB; fi
  timeout -k 10 300 python3 tools/bench_reads.py --steps 10 --warm 6 --no-check > $O/reads_$v.json 2> $O/reads_$v.err || { tail -5 $O/reads_$v.err; exit 1; }
  timeout -k 10 300 python3 tools/bench_chain.py --no-unfused --steps 12 --warm 6 > $O/chain_$v.json 2> $O/chain_$v.err || { tail -5 $O/chain_$v.err; exit 1; }
  [ -n "$NO_CALL" ] || timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-chain --no-reads --steps 20 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python3 - <<PY
import json, os
r = json.loads(open("$O/reads_$v.json").read().strip().splitlines()[-1])
ch = json.loads(open("$O/chain_$v.json").read().strip().splitlines()[-1])
s = "%-10s reads %.3f ms (min %.3f)  accumulate %.3f ms  chain %.3f (dev %.3f)" % ("$v", r["reads_chain"]["device_ms_avg"], r["reads_chain"]["device_ms_min"], r["accumulate"]["device_ms_avg"], ch["fused_ms"], ch["fused_device_ms_last_window"])
if os.path.exists("$O/bench_$v.json") and not "$NO_CALL":
    b = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
    s += "  call %.3f ms (min %.3f)" % (b["roofline"]["kernel_ms_avg"], b["roofline"]["kernel_ms_min"])
print(s)
PY
done
done
