#!/bin/bash
# Round 4 working script (GPU box): parity tests of the chain kernels, then timings of the reads-in chain and the pile-up-in
# chain under different run caps (BSC_CHAIN_RUN_CAP: 1 = round 3's layout, tiles of 60 interleaved over the waves).
# usage: bash tools/r04_step.sh <tag> [caps...]
set -e
TAG=$1; shift
CAPS=${@:-"32"}
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
[ -n "$SKIP_TESTS" ] || timeout -k 10 900 python3 -m pytest tests/test_gpu_accumulate.py tests/test_gpu_chain.py tests/test_gpu_reads_chain.py tests/test_gpu_records.py tests/test_gpu_fullsize.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -3 $O/pytest.txt
for v in ${VARIANTS:-main}; do
for c in $CAPS; do
  if [ $v != main ]; then export BSCALL_AMD_LIB=$ROOT/bs_call_amd/lib/variants/lib_$v.so; else unset BSCALL_AMD_LIB; fi
  BSC_CHAIN_RUN_CAP=$c timeout -k 10 300 python3 tools/bench_reads.py --steps 10 --warm 5 --no-check > $O/reads_cap$c.json 2> $O/reads_cap$c.err || { tail -5 $O/reads_cap$c.err; exit 1; }
  BSC_CHAIN_RUN_CAP=$c timeout -k 10 300 python3 tools/bench_chain.py --no-unfused --steps 10 --warm 5 > $O/chain_cap$c.json 2> $O/chain_cap$c.err || { tail -5 $O/chain_cap$c.err; exit 1; }
  python3 - <<PY
import json
r = json.loads(open("$O/reads_cap$c.json").read().strip().splitlines()[-1])
ch = json.loads(open("$O/chain_cap$c.json").read().strip().splitlines()[-1])
print("$v cap $c: reads chain %.3f ms (min %.3f)  accumulate %.3f ms   chain %s" % (r["reads_chain"]["device_ms_avg"], r["reads_chain"]["device_ms_min"], r["accumulate"]["device_ms_avg"], {k: ch[k] for k in ch if "ms" in k}))
PY
done
done
