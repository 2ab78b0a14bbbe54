# FETCH_SIZE / WRITE_SIZE passes of the fused chain kernel alone (the chain half of tools/prof_final.sh), into an existing
# gpurun_out/prof_<tag> directory; then profiles/traffic.json is rewritten.  usage (GPU box): bash tools/pmc_chain_traffic.sh <tag>
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $ROOT/$OUT
cd /tmp
rm -rf $ROOT/$OUT/chain_fetch $ROOT/$OUT/chain_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/chain_fetch -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2 > /dev/null 2> $ROOT/$OUT/chain_fetch.err || { tail -5 $ROOT/$OUT/chain_fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/chain_write -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2 > /dev/null 2> $ROOT/$OUT/chain_write.err || { tail -5 $ROOT/$OUT/chain_write.err; exit 1; }
cd $ROOT
find $OUT/chain_fetch $OUT/chain_write -name "*counter_collection.csv" | head
