#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of the bench command, then PMC passes for HBM traffic.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain "$@" > $ROOT/$OUT/bench_trace.json 2> $ROOT/$OUT/trace.err || { tail -5 $ROOT/$OUT/trace.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --steps 2 --warmup 1 > $ROOT/$OUT/bench_pmc1.json 2> $ROOT/$OUT/pmc1.err || { tail -5 $ROOT/$OUT/pmc1.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --steps 2 --warmup 1 > $ROOT/$OUT/bench_pmc2.json 2> $ROOT/$OUT/pmc2.err || { tail -5 $ROOT/$OUT/pmc2.err; exit 1; }
cd $ROOT
find $OUT -name "*.csv" | head -20
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
