# Everything the final numbers of a round come from, in one gpurun call: usage (on the GPU box): bash tools/prof_final.sh <tag>
#   1. bsc_call_kernel: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of bench.py   (tools/profile_bench.sh)
#   2. bsc_chain_kernel_t: kernel-trace stats, FETCH_SIZE / WRITE_SIZE passes and SQ counters of tools/bench_chain.py
#   3. profiles/traffic.json from 1 + 2 (tools/make_traffic_json.py), then the plain bench line and the configs[2] / [4] lines
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
bash tools/profile_bench.sh $TAG --steps 10 --warmup 3 > gpurun_out/prof_$TAG.log 2>&1 || { tail -20 gpurun_out/prof_$TAG.log; exit 1; }
tail -8 gpurun_out/prof_$TAG.log
OUT=gpurun_out/prof_$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/chain_trace -- python3 $ROOT/tools/bench_chain.py --steps 8 > $ROOT/$OUT/chain_bench.json 2> $ROOT/$OUT/chain_trace.err || { tail -5 $ROOT/$OUT/chain_trace.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/chain_fetch -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2 > /dev/null 2> $ROOT/$OUT/chain_fetch.err || { tail -5 $ROOT/$OUT/chain_fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/chain_write -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2 > /dev/null 2> $ROOT/$OUT/chain_write.err || { tail -5 $ROOT/$OUT/chain_write.err; exit 1; }
cd $ROOT
python3 tools/make_traffic_json.py $OUT
bash tools/pmc_sq.sh $TAG > gpurun_out/sq_$TAG.txt 2>&1 || { tail -20 gpurun_out/sq_$TAG.txt; exit 1; }
bash tools/pmc_chain.sh $TAG > gpurun_out/sqc_$TAG.txt 2>&1 || { tail -20 gpurun_out/sqc_$TAG.txt; exit 1; }
find $OUT/chain_trace -name "*kernel_stats.csv" | xargs cat | cut -c1-200 > gpurun_out/chain_stats_$TAG.csv
cat $OUT/chain_bench.json
timeout -k 10 500 python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
cat gpurun_out/bench_$TAG.json
timeout -k 10 300 python3 bench.py --config 3 --rank-of 8 --steps 3 --warmup 1 > gpurun_out/cfg3_$TAG.json 2> gpurun_out/cfg3_$TAG.err
timeout -k 10 300 python3 bench.py --config 3 --rank-of 8 --steps 3 --warmup 1 --window 16711680 > gpurun_out/cfg3w16_$TAG.json 2>> gpurun_out/cfg3_$TAG.err
timeout -k 10 300 python3 bench.py --config 3 --rank-of 8 --steps 3 --warmup 1 --dbsnp > gpurun_out/cfg5_$TAG.json 2>> gpurun_out/cfg3_$TAG.err
python3 - <<PY
import json
for f in ("cfg3_$TAG", "cfg3w16_$TAG", "cfg5_$TAG"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e9, 3), "G positions/s", round(d["ms_per_step"], 3), "ms", d["config"]["windows_per_step"], "windows")
PY
