#!/bin/bash
# Copies what tools/prof_r05.sh left under gpurun_out/<tag>/ (both parts) into profiles/ under the round's names (run here,
# after the gpurun calls have merged their output back).  usage: bash tools/collect_r04.sh <tag>
set -e
O=gpurun_out/$1
P=profiles
hdr() { if head -1 $1 | grep -q "^#"; then sed -i "1s|.*|# $2|" $1; else sed -i "1i # $2" $1; fi; }
cp $O/call_kernel_timed.txt $P/r05_call_kernel_timed.txt
cp $(ls $O/call_trace/*/*_kernel_stats.csv | head -1) $P/r05_call_kernel_stats.csv
cp $O/chain_kernel_timed.txt $P/r05_chain_kernel_timed.txt
cp $(ls $O/chain_trace/*/*_kernel_stats.csv | head -1) $P/r05_chain_kernel_stats.csv
for c in 30x 200x; do
  [ -f $O/reads_kernels_timed_$c.txt ] || continue
  cp $O/reads_kernels_timed_$c.txt $P/r05_reads_kernels_timed_$c.txt
  cp $(ls $O/reads_trace_$c/*/*_kernel_stats.csv | head -1) $P/r05_reads_kernel_stats_$c.csv
  cp $O/reads_$c.json $P/r05_reads_$c.json
  [ -f $O/reads_one_kernel_$c.json ] && cp $O/reads_one_kernel_$c.json $P/r05_reads_one_kernel_$c.json
done
cp $O/call_sq_counters_30x.txt $P/r05_call_sq_counters_30x.txt
cp $O/chain_sq_counters_30x.txt $P/r05_chain_sq_counters_30x.txt
cp $O/reads_chain_sq_counters_30x.txt $P/r05_reads_chain_sq_counters_30x.txt
cp $O/accsum_sq_counters_30x.txt $P/r05_acc_sq_counters_30x.txt
hdr $P/r05_call_sq_counters_30x.txt "bsc_call_kernel_t<true>, 50 M positions at 30x, mean per launch (tools/pmc_sq.sh)"
hdr $P/r05_chain_sq_counters_30x.txt "bsc_chain_kernel_t<true, false, false, false> (pile-up in), 50 M positions at 30x with statistics, mean per launch (tools/pmc_chain.sh)"
hdr $P/r05_reads_chain_sq_counters_30x.txt "bsc_chain_kernel_t<true, false, false, true> (summary in: the chain kernel of the reads path), one block of 50 M positions at 30x, mean per launch (tools/pmc_kernel.sh ... tools/bench_reads.py)"
cp $O/traffic.json $P/traffic.json
cp $O/valu.json $P/valu.json
for f in bench bench_cfg4_10Mb_200x bench_cfg1_1Mb_10x cfg3_rank0of8 cfg5_rank0of8; do
  [ -f $O/$f.json ] && cp $O/$f.json $P/r05_${f#bench_}.json
done
[ -f $O/small_blocks.txt ] && { grep -v amdgpu.ids $O/small_blocks.txt > $P/r05_small_blocks.txt; cp $O/small_blocks.json $P/r05_small_blocks.json; hdr $P/r05_small_blocks.txt "tools/bench_small_blocks.py: host buffers in, packed records out, wall time (best of 5 passes); batches of >= 1 M positions"; }
[ -f $O/bam2bcf.json ] && cp $O/bam2bcf.json $P/r05_bam2bcf.json
sed -i "s#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#; s#/root/repo/gpurun_out/#gpurun_out/#" $P/r05_*_timed.txt
python3 tools/make_resources.py > /dev/null
echo collected $1
