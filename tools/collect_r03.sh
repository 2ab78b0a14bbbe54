#!/bin/bash
# Copies what tools/prof_r03.sh left under gpurun_out/<tag>/ into profiles/ under the round's names (run here, after the
# gpurun call has merged its output back).  usage: bash tools/collect_r03.sh <tag>
set -e
O=gpurun_out/$1
P=profiles
cp $O/call_kernel_timed.txt $P/r03_call_kernel_timed.txt
cp $(ls $O/call_trace/*/*_kernel_stats.csv | head -1) $P/r03_call_kernel_stats.csv
cp $O/chain_kernel_timed.txt $P/r03_chain_kernel_timed.txt
cp $(ls $O/chain_trace/*/*_kernel_stats.csv | head -1) $P/r03_chain_kernel_stats.csv
for c in 30x 200x; do
  cp $O/reads_kernels_timed_$c.txt $P/r03_reads_kernels_timed_$c.txt
  cp $(ls $O/reads_trace_$c/*/*_kernel_stats.csv | head -1) $P/r03_reads_kernel_stats_$c.csv
  cp $O/reads_$c.json $P/r03_reads_$c.json
  cp $O/call_sq_counters_$c.txt $P/r03_call_sq_counters_$c.txt
  cp $O/chain_sq_counters_$c.txt $P/r03_chain_sq_counters_$c.txt
  cp $O/reads_chain_sq_counters_$c.txt $P/r03_reads_chain_sq_counters_$c.txt
  cp $O/accumulate_sq_counters_$c.txt $P/r03_acc_sq_counters_$c.txt
done
cp $O/bench.json $P/r03_bench.json
cp $O/bench_cfg4_10Mb_200x.json $P/r03_cfg4_10Mb_200x.json
cp $O/bench_cfg1_1Mb_10x.json $P/r03_cfg1_1Mb_10x.json
cp $O/cfg3_rank0of8.json $P/r03_cfg3_rank0of8.json
cp $O/cfg5_rank0of8.json $P/r03_cfg5_rank0of8.json
hdr() { # hdr <file> <header line>: one descriptive first line per counters file
  if head -1 $1 | grep -q "^#"; then sed -i "1s|.*|# $2|" $1; else sed -i "1i # $2" $1; fi
}
hdr $P/r03_call_sq_counters_30x.txt "bsc_call_kernel_t<true>, 50 M positions at 30x, mean per launch (tools/pmc_sq.sh)"
hdr $P/r03_call_sq_counters_200x.txt "bsc_call_kernel_t<true>, 10 M positions at 200x, mean per launch (tools/pmc_sq.sh --sites 10000000 --coverage 200)"
hdr $P/r03_chain_sq_counters_30x.txt "bsc_chain_kernel_t<true, false>, 50 M positions at 30x with statistics, mean per launch (tools/pmc_chain.sh)"
hdr $P/r03_chain_sq_counters_200x.txt "bsc_chain_kernel_t<true, false>, 10 M positions at 200x with statistics, mean per launch (tools/pmc_chain.sh --sites 10000000 --coverage 200)"
hdr $P/r03_reads_chain_sq_counters_30x.txt "bsc_chain_kernel_t<true, true> (reads-in chain), one block of 50 M positions at 30x, mean per launch (tools/pmc_kernel.sh ... tools/bench_reads.py)"
hdr $P/r03_reads_chain_sq_counters_200x.txt "bsc_chain_kernel_t<true, true> (reads-in chain), one block of 10 M positions at 200x, mean per launch (tools/pmc_kernel.sh ... tools/bench_reads.py --sites 10000000 --coverage 200)"
hdr $P/r03_acc_sq_counters_30x.txt "bsc_accumulate_kernel, one block of 50 M positions at 30x, mean per launch (tools/pmc_kernel.sh ... tools/bench_reads.py --no-chain)"
hdr $P/r03_acc_sq_counters_200x.txt "bsc_accumulate_kernel, one block of 10 M positions at 200x, mean per launch (tools/pmc_kernel.sh ... tools/bench_reads.py --no-chain --sites 10000000 --coverage 200)"
python3 -c "import json,sys; json.dump(json.load(open('$O/traffic.stdout')), open('$P/traffic.json','w'), indent=1)"
sed -i "s#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#" $P/r03_*_timed.txt
echo collected $1
