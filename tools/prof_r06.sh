#!/bin/bash
# Round 6's numbers of record, in gpurun calls of a few minutes each (GPU box).  usage: bash tools/prof_r06.sh <tag> a|b|c
#   a  tools/prof_r05.sh a: calling kernel, pile-up-in chain, reads: kernel traces, FETCH / WRITE passes, SQ counters; traffic.json, valu.json
#   b  tools/prof_r05.sh b without the file-to-file run: reads at 200x, small blocks, the plain bench lines, configs[2] / [4] rank 0 of 8
#   c  round 6's own kernels:
#      1. read pre-processing (tools/bench_prep.py, without and with the read profile): kernel trace, FETCH / WRITE / SQ_INSTS_VALU passes
#      2. the BCF encoder over the chain's arrays, sized from the length bytes (tools/bench_sites_bcf.py): kernel trace, FETCH / WRITE passes
#      3. profiles/traffic.json, valu.json entries prep / prep_profile / bcf_sites (tools/make_r06_json.py)
#      4. the device BAM reader's kernels: kernel trace of integration/bam2bcf over a 50 Mb / 30x file; tools/bench_bam2bcf_big.py
#      5. the glue's protocols end to end (tools/r06_glue.sh); bench.py
set -e
TAG=$1
PART=${2:-c}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
if [ $PART = a ] || [ $PART = b ]; then
  if [ $PART = b ]; then sed -i 's#^timeout -k 10 900 python3 tools/bench_bam2bcf.py.*#true#' tools/prof_r05.sh; fi
  exec bash tools/prof_r05.sh $TAG $PART
fi
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
prof() { # prof <name> <rocprof args...> -- program args...
  local out=$1; shift
  (cd /tmp && timeout -k 10 400 rocprofv3 "$@" > $O/$out.out 2> $O/$out.err) || { tail -5 $O/$out.err; exit 1; }
}
# 1
prof prep_trace --kernel-trace --stats --output-format csv -d $O/prep_trace -- python3 $ROOT/tools/bench_prep.py --steps 8
python3 tools/kstats_timed.py $O/prep_trace 2 > $O/prep_kernels_timed.txt; cat $O/prep_kernels_timed.txt
prof prepp_trace --kernel-trace --stats --output-format csv -d $O/prepp_trace -- python3 $ROOT/tools/bench_prep.py --profile --steps 8
python3 tools/kstats_timed.py $O/prepp_trace 2 > $O/prep_profile_kernels_timed.txt; cat $O/prep_profile_kernels_timed.txt
for c in FETCH_SIZE WRITE_SIZE; do
  prof prep_$c --kernel-trace --pmc $c --output-format csv -d $O/prep_$c -- python3 $ROOT/tools/bench_prep.py --steps 2
  prof prepp_$c --kernel-trace --pmc $c --output-format csv -d $O/prepp_$c -- python3 $ROOT/tools/bench_prep.py --profile --steps 2
done
prof prep_SQ --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/prep_SQ -- python3 $ROOT/tools/bench_prep.py --steps 2
prof prepp_SQ --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/prepp_SQ -- python3 $ROOT/tools/bench_prep.py --profile --steps 2
# 2
prof sites_trace --kernel-trace --stats --output-format csv -d $O/sites_trace -- python3 $ROOT/tools/bench_sites_bcf.py --steps 8
python3 tools/kstats_timed.py $O/sites_trace 2 bsc_bcf > $O/bcf_sites_kernels_timed.txt; cat $O/bcf_sites_kernels_timed.txt
for c in FETCH_SIZE WRITE_SIZE; do
  prof sites_$c --kernel-trace --pmc $c --output-format csv -d $O/sites_$c -- python3 $ROOT/tools/bench_sites_bcf.py --steps 2
done
# 3
python3 tools/make_r06_json.py $O | tee $O/r06_json.txt
cp profiles/traffic.json $O/traffic.json
cp profiles/valu.json $O/valu.json
# 4
T=/tmp/r06_bam
mkdir -p $T
[ -x $T/make_wgbs_bam ] || gcc -O2 -o $T/make_wgbs_bam tools/make_wgbs_bam.c -lz -lpthread -lm
$T/make_wgbs_bam $T/in.bam $T/ref.fa 50000000 30 7 16 1 > $O/bam_generated.txt
(cd /tmp && BAM2BCF_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bam2bcf_trace -- $ROOT/bs_call_amd/lib/bam2bcf $T/in.bam $T/ref.fa $T/out.bcf $T/rep.json > $O/bam2bcf_trace.out 2> $O/bam2bcf_trace.err) || { tail -5 $O/bam2bcf_trace.err; exit 1; }
python3 tools/kstats.py $(ls $O/bam2bcf_trace/*/*_kernel_stats.csv | head -1) 40 > $O/bam2bcf_kernels.txt; cat $O/bam2bcf_kernels.txt
rm -rf $T
timeout -k 10 400 python3 tools/bench_bam2bcf_big.py 50000000 30 $O/bam2bcf_50Mb.json > $O/bam2bcf_50Mb.log 2>&1 || { tail -5 $O/bam2bcf_50Mb.log; exit 1; }
tail -1 $O/bam2bcf_50Mb.log | cut -c1-400
timeout -k 10 400 python3 tools/bench_bam2bcf_big.py 50000000 30 $O/bam2bcf_50Mb_8contigs.json 8 > $O/bam2bcf_50Mb_8contigs.log 2>&1 || { tail -5 $O/bam2bcf_50Mb_8contigs.log; exit 1; }
grep "^device_reader" $O/bam2bcf_50Mb_8contigs.log | cut -c1-330
# 5
tools/r06_glue.sh > $O/glue_demo.txt 2>&1 || { tail -5 $O/glue_demo.txt; exit 1; }
grep "end to end" $O/glue_demo.txt
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo part c done
