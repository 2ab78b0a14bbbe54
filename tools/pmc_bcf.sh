#!/bin/bash
# HBM traffic of the BCF encoder's kernels (GPU box): FETCH_SIZE and WRITE_SIZE passes (separate runs, kernel trace only beside them) of
# tools/bench_bcf.py at configs[1] size, then tools/make_bcf_traffic.py -> profiles/traffic.json["bcf"], profiles-ready text on stdout.
# usage: bash tools/pmc_bcf.sh <tag>
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $ROOT/tools/bench_bcf.py --sites 50000000 --steps 2 --host-sample 1000 > $O/pmc_$c.json 2> $O/pmc_$c.err) || { tail -5 $O/pmc_$c.err; exit 1; }
done
python3 tools/make_bcf_traffic.py $O | tee $O/bcf_traffic.txt
cp profiles/traffic.json $O/traffic.json
