#!/bin/bash
# Everything the final numbers of round 5 come from, in TWO gpurun calls (each within the 20-minute limit).
# usage (on the GPU box): bash tools/prof_r05.sh <tag> a|b
#   a  1. calling kernel: kernel trace of bench.py (20 timed launches; warm-ups reported apart), FETCH_SIZE / WRITE_SIZE passes,
#         SQ counters at 30x
#      2. pile-up-in chain: kernel trace + FETCH / WRITE passes + SQ counters of tools/bench_chain.py
#      3. reads: kernel trace of tools/bench_reads.py at configs[1] size (50 Mb, 30x), FETCH / WRITE passes, SQ counters of the
#         two kernels of the reads path and of the stand-alone accumulate kernel
#      4. profiles/traffic.json, profiles/valu.json
#   b  5. reads at configs[3] size (10 Mb, 200x): kernel trace; the one-kernel form at both sizes (bsc_set_reads_fused)
#      6. small blocks (tools/bench_small_blocks.py)
#      7. the plain bench lines (30x, 200x, 10x) and the configs[2] / [4] rank-0-of-8 lines
set -e
TAG=$1
PART=${2:-a}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O/prof
prof() { # prof <outdir> <rocprof args...> -- script args...   (runs from /tmp, program directly after --)
  local out=$1; shift
  (cd /tmp && rocprofv3 "$@" > $O/$out.stdout 2> $O/$out.err) || { tail -5 $O/$out.err; exit 1; }
}
if [ $PART = a ]; then
# 1
prof call_trace --kernel-trace --stats --output-format csv -d $O/call_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --no-reads --steps 20 --warmup 5
python3 tools/kstats_timed.py $O/call_trace 5 bsc_ > $O/call_kernel_timed.txt; cat $O/call_kernel_timed.txt
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --no-reads --steps 2 --warmup 1
prof pmc_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-chain --no-reads --steps 2 --warmup 1
bash tools/pmc_sq.sh ${TAG}_30x --no-reads > $O/call_sq_counters_30x.txt 2>&1 || { tail $O/call_sq_counters_30x.txt; exit 1; }
# 2
prof chain_trace --kernel-trace --stats --output-format csv -d $O/chain_trace -- python3 $ROOT/tools/bench_chain.py --steps 20 --no-unfused --warm 8
python3 tools/kstats_timed.py $O/chain_trace 8 bsc_chain > $O/chain_kernel_timed.txt; cat $O/chain_kernel_timed.txt
prof chain_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/chain_fetch -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2
prof chain_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/chain_write -- python3 $ROOT/tools/bench_chain.py --no-unfused --steps 2
bash tools/pmc_chain.sh ${TAG}_30x --steps 2 > $O/chain_sq_counters_30x.txt 2>&1 || { tail $O/chain_sq_counters_30x.txt; exit 1; }
# 3
prof reads_trace_30x --kernel-trace --stats --output-format csv -d $O/reads_trace_30x -- python3 $ROOT/tools/bench_reads.py --steps 10 --warm 10
cp $O/reads_trace_30x.stdout $O/reads_30x.json
python3 tools/kstats_timed.py $O/reads_trace_30x 10 > $O/reads_kernels_timed_30x.txt; cat $O/reads_kernels_timed_30x.txt
bash tools/pmc_kernel.sh ${TAG}_rc30 "bsc_chain_kernel_t<true, false, false, true>" tools/bench_reads.py --steps 2 --no-check > $O/reads_chain_sq_counters_30x.txt 2>&1 || { tail $O/reads_chain_sq_counters_30x.txt; exit 1; }
python3 - <<PY > $O/accsum_sq_counters_30x.txt
import csv, glob
for kern in ("bsc_accumulate_kernel_t<true>", "bsc_accumulate_kernel_t<false>"):
    agg = {}
    for f in glob.glob("gpurun_out/sq_${TAG}_rc30/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]: continue
            k = (r["Counter_Name"], r["Dispatch_Id"]); agg[k] = agg.get(k, 0) + float(r["Counter_Value"])
    per = {}
    for (c, _), v in agg.items(): per.setdefault(c, []).append(v)
    print("# %s, one block of 50 M positions at 30x, mean per launch (the passes of tools/pmc_kernel.sh ... tools/bench_reads.py)" % kern)
    for c, v in sorted(per.items()): print("%-24s %.4g" % (c, sum(v) / len(v)))
PY
prof reads_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/reads_fetch -- python3 $ROOT/tools/bench_reads.py --steps 2 --no-check
prof reads_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/reads_write -- python3 $ROOT/tools/bench_reads.py --steps 2 --no-check
# 4
python3 tools/make_traffic_json.py $O/prof > $O/traffic.stdout
cp profiles/traffic.json $O/traffic.json
python3 tools/make_valu_json.py --call gpurun_out/sq_${TAG}_30x --chain gpurun_out/sqc_${TAG}_30x --reads gpurun_out/sq_${TAG}_rc30 --acc gpurun_out/sq_${TAG}_rc30 > $O/valu.stdout
cp profiles/valu.json $O/valu.json
echo part a done
else
# 5
prof reads_trace_200x --kernel-trace --stats --output-format csv -d $O/reads_trace_200x -- python3 $ROOT/tools/bench_reads.py --sites 10000000 --coverage 200 --steps 10 --warm 10
cp $O/reads_trace_200x.stdout $O/reads_200x.json
python3 tools/kstats_timed.py $O/reads_trace_200x 10 > $O/reads_kernels_timed_200x.txt; cat $O/reads_kernels_timed_200x.txt
python3 tools/bench_reads.py --one-kernel --steps 10 --warm 10 --no-check > $O/reads_one_kernel_30x.json 2> $O/reads_one_kernel.err
python3 tools/bench_reads.py --one-kernel --sites 10000000 --coverage 200 --steps 10 --warm 10 --no-check > $O/reads_one_kernel_200x.json 2>> $O/reads_one_kernel.err
# 6
timeout -k 10 600 python3 tools/bench_small_blocks.py > $O/small_blocks.json 2> $O/small_blocks.txt; cat $O/small_blocks.txt
# 6b file to file: BAM + FASTA -> BCF + report (pre-processing on the device, and on the host as in round 4)
timeout -k 10 900 python3 tools/bench_bam2bcf.py 2000000 > $O/bam2bcf.json 2> $O/bam2bcf.err || { tail -5 $O/bam2bcf.err; exit 1; }
# 7
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --sites 10000000 --coverage 200 --no-cpu-baseline --warmup 40 > $O/bench_cfg4_10Mb_200x.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --sites 1000000 --coverage 10 --no-cpu-baseline --warmup 200 > $O/bench_cfg1_1Mb_10x.json 2>> $O/bench.err
timeout -k 10 300 python3 bench.py --config 3 --rank-of 8 --steps 3 --warmup 1 > $O/cfg3_rank0of8.json 2> $O/cfg3.err
timeout -k 10 300 python3 bench.py --config 3 --rank-of 8 --steps 3 --warmup 1 --dbsnp > $O/cfg5_rank0of8.json 2>> $O/cfg3.err
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("bench:", round(d["value"] / 1e9, 3), "G positions/s; call frac", round(d["roofline"]["frac"], 4), "valu", d["roofline"].get("valu") and round(d["roofline"]["valu"]["frac"], 3), "traffic", d["roofline"]["traffic"])
for k in ("roofline_chain", "roofline_accumulate", "roofline_reads"):
    r = d[k]; print(k, round(r.get("kernel_ms_avg", r.get("stage_ms_avg")), 3), "ms  hbm", round(r["frac"], 4), " valu", r.get("valu") and round(r["valu"]["frac"], 3), " traffic", r["traffic"])
for f in ("cfg3_rank0of8", "cfg5_rank0of8"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e9, 3), "G positions/s", round(d["ms_per_step"], 3), "ms")
PY
echo part b done
fi
