#!/bin/bash
# A/B of library build variants on the pile-up-in chain (tools/bench_chain.py), alternating, un-profiled.
# usage (GPU box): bash tools/ab_chain.sh <tag> <variant names... | main>
set -e
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v != main ]; then export BSCALL_AMD_LIB=$ROOT/bs_call_amd/lib/variants/lib_$v.so; else unset BSCALL_AMD_LIB; fi
  timeout -k 10 300 python3 $ROOT/tools/bench_chain.py --no-unfused --steps 20 --warm 8 > $O/$v.$rep.json 2> $O/$v.err || { tail -5 $O/$v.err; exit 1; }
  python3 -c "
import json
d = json.loads(open('$O/$v.$rep.json').read().strip().splitlines()[-1])
print('$v', {k: round(d[k], 3) for k in d if 'ms' in k})"
done
done
