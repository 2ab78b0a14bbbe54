#!/bin/bash
# Everything the final numbers of round 3 come from, in one gpurun call.  usage (on the GPU box): bash tools/prof_r03.sh <tag>
#   1. calling kernel: kernel trace of bench.py (20 timed launches; warm-ups reported apart), FETCH_SIZE / WRITE_SIZE passes
#   2. pile-up-in chain: kernel trace + FETCH / WRITE passes + SQ counters of tools/bench_chain.py
#   3. reads-in: kernel trace of tools/bench_reads.py at configs[1] (50 Mb, 30x) and configs[3] (10 Mb, 200x) sizes, FETCH /
#      WRITE passes and SQ counters of bsc_chain_kernel_t<true, true> and bsc_accumulate_kernel at both sizes
#   4. SQ counters of the calling and chain kernels at 200x
#   5. profiles/traffic.json, then the plain bench lines (30x, 200x) and the configs[2] / [4] rank-0-of-8 lines
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
prof() { # prof <outdir> <rocprof args...> -- script args...   (runs from /tmp, program directly after --)
  local out=$1; shift
  (cd /tmp && rocprofv3 "$@" > $O/$out.stdout 2> $O/$out.err) || { tail -5 $O/$out.err; exit 1; }
}
# 1
prof call_trace --kernel-trace --stats --output-format csv -d $O/call_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --no-reads --steps 20 --warmup 5
python3 tools/kstats_timed.py $O/call_trace 5 bsc_ > $O/call_kernel_timed.txt; cat $O/call_kernel_timed.txt
mkdir -p $O/prof; 
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --no-reads --steps 2 --warmup 1
prof pmc_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --no-reads --steps 2 --warmup 1
# 2
prof chain_trace --kernel-trace --stats --output-format csv -d $O/chain_trace -- python3 $ROOT/tools/bench_chain.py --steps 20 --no-unfused --warm 8
python3 tools/kstats_timed.py $O/chain_trace 8 bsc_chain > $O/chain_kernel_timed.txt; cat $O/chain_kernel_timed.txt
prof chain_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/chain_fetch -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2
prof chain_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/chain_write -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2
bash tools/pmc_chain.sh ${TAG}_30x --steps 2 > $O/chain_sq_counters_30x.txt 2>&1 || { tail $O/chain_sq_counters_30x.txt; exit 1; }
# 3
for CFG in "50000000 30" "10000000 200"; do
  set -- $CFG
  prof reads_trace_$2x --kernel-trace --stats --output-format csv -d $O/reads_trace_$2x -- python3 $ROOT/tools/bench_reads.py --sites $1 --coverage $2 --steps 10 --warm 10
  cp $O/reads_trace_$2x.stdout $O/reads_$2x.json
  python3 tools/kstats_timed.py $O/reads_trace_$2x 10 > $O/reads_kernels_timed_$2x.txt; cat $O/reads_kernels_timed_$2x.txt
  bash tools/pmc_kernel.sh ${TAG}_rc$2 "bsc_chain_kernel_t<true, true>" tools/bench_reads.py --sites $1 --coverage $2 --steps 2 --no-check > $O/reads_chain_sq_counters_$2x.txt 2>&1 || { tail $O/reads_chain_sq_counters_$2x.txt; exit 1; }
  bash tools/pmc_kernel.sh ${TAG}_acc$2 bsc_accumulate_kernel tools/bench_reads.py --sites $1 --coverage $2 --steps 2 --no-check --no-chain > $O/accumulate_sq_counters_$2x.txt 2>&1 || { tail $O/accumulate_sq_counters_$2x.txt; exit 1; }
done
prof reads_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/reads_fetch -- python3 $ROOT/tools/bench_reads.py --steps 2 --no-check
prof reads_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/reads_write -- python3 $ROOT/tools/bench_reads.py --steps 2 --no-check
# 4
bash tools/pmc_sq.sh ${TAG}_200x --sites 10000000 --coverage 200 --no-reads > $O/call_sq_counters_200x.txt 2>&1 || { tail $O/call_sq_counters_200x.txt; exit 1; }
bash tools/pmc_sq.sh ${TAG}_30x --no-reads > $O/call_sq_counters_30x.txt 2>&1 || { tail $O/call_sq_counters_30x.txt; exit 1; }
bash tools/pmc_chain.sh ${TAG}_200x --steps 2 --sites 10000000 --coverage 200 > $O/chain_sq_counters_200x.txt 2>&1 || { tail $O/chain_sq_counters_200x.txt; exit 1; }
# 5
python3 tools/make_traffic_json.py $O/prof > $O/traffic.stdout
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --sites 10000000 --coverage 200 --no-cpu-baseline --warmup 40 > $O/bench_cfg4_10Mb_200x.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --sites 1000000 --coverage 10 --no-cpu-baseline --warmup 200 > $O/bench_cfg1_1Mb_10x.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --config 3 --rank-of 8 --steps 3 --warmup 1 > $O/cfg3_rank0of8.json 2> $O/cfg3.err
timeout -k 10 300 python3 bench.py --config 3 --rank-of 8 --steps 3 --warmup 1 --dbsnp > $O/cfg5_rank0of8.json 2>> $O/cfg3.err
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("bench:", round(d["value"] / 1e9, 3), "G positions/s; call frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"])
for k in ("roofline_chain", "roofline_accumulate", "roofline_reads"):
    r = d[k]; print(k, round(r.get("kernel_ms_avg", r.get("stage_ms_avg")), 3), "ms", round(r["frac"], 4), "traffic", r["traffic"])
for f in ("cfg3_rank0of8", "cfg5_rank0of8"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e9, 3), "G positions/s", round(d["ms_per_step"], 3), "ms")
PY
