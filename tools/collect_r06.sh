#!/bin/bash
# Copies what tools/prof_r06.sh left under gpurun_out/ into profiles/ under the round's names (run here, after the gpurun calls have merged
# their output back).  usage: bash tools/collect_r06.sh <tag of part c> [<tag of parts a / b>]
set -e
P=profiles
hdr() { if head -1 $1 | grep -q "^#"; then sed -i "1s|.*|# $2|" $1; else sed -i "1i # $2" $1; fi; }
if [ -n "$1" ] && [ -d gpurun_out/$1 ]; then
  O=gpurun_out/$1
  cp $O/prep_kernels_timed.txt $P/r06_prep_kernels_timed.txt
  cp $O/prep_profile_kernels_timed.txt $P/r06_prep_profile_kernels_timed.txt
  cp $O/bcf_sites_kernels_timed.txt $P/r06_bcf_sites_kernels_timed.txt
  cp $O/r06_json.txt $P/r06_prep_bcf_traffic.txt
  python3 - $O <<'PY'
import json, sys
for name, keys in (("traffic.json", ("prep", "prep_profile", "bcf_sites")), ("valu.json", ("prep", "prep_profile"))):
    cur, new = json.load(open("profiles/" + name)), json.load(open(sys.argv[1] + "/" + name))
    for k in keys:
        if k in new: cur[k] = new[k]
    json.dump(cur, open("profiles/" + name, "w"), indent=1)
PY
  hdr $P/r06_prep_bcf_traffic.txt "tools/make_r06_json.py: HBM bytes ((2 x FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes) and VALU wave-instructions per call of the round's legs, 50 M positions at 30x; the entries of traffic.json / valu.json"
  { echo "# rocprofv3 --kernel-trace --stats -- bs_call_amd/lib/bam2bcf over a 50 Mb / 30x BAM (15 M alignments, one contig): every kernel of the run, file to file"; cat $O/bam2bcf_kernels.txt; grep -h "^{" $O/bam2bcf_trace.err | tail -1; } > $P/r06_bam2bcf_kernels.txt
  cp $O/bam2bcf_50Mb.json $P/r06_bam2bcf_50Mb.json
  cp $O/glue_demo.txt $P/r06_glue_demo.txt
  cp $O/bench.json $P/r06_bench_c.json
  for t in prep prepp; do
    python3 - $O $t <<'PY' > $P/r06_${t/prepp/prep_profile}_sq_counters.txt
import csv, glob, sys
o, t = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob("%s/%s_SQ/**/*counter_collection.csv" % (o, t), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0]
        if not kn.startswith("bsc_prep"): continue
        k = (kn, r["Counter_Name"], r["Dispatch_Id"]); agg[k] = agg.get(k, 0) + float(r["Counter_Value"])
per = {}
for (kn, c, _), v in agg.items(): per.setdefault((kn, c), []).append(v)
print("# tools/bench_prep.py %s--steps 2 under rocprofv3 --pmc (tools/prof_r06.sh): SQ counters of the pre-processing's kernels, mean per launch, 50 M positions at 30x" % ("--profile " if t == "prepp" else ""))
for (kn, c), v in sorted(per.items()): print("%-28s %-22s %.4g" % (kn, c, sum(v) / len(v)))
PY
  done
fi
if [ -n "$2" ] && [ -d gpurun_out/$2 ]; then
  O=gpurun_out/$2
  [ -f $O/call_kernel_timed.txt ] && { cp $O/call_kernel_timed.txt $P/r06_call_kernel_timed.txt; cp $(ls $O/call_trace/*/*_kernel_stats.csv | head -1) $P/r06_call_kernel_stats.csv; }
  [ -f $O/chain_kernel_timed.txt ] && { cp $O/chain_kernel_timed.txt $P/r06_chain_kernel_timed.txt; cp $(ls $O/chain_trace/*/*_kernel_stats.csv | head -1) $P/r06_chain_kernel_stats.csv; }
  for c in 30x 200x; do
    [ -f $O/reads_kernels_timed_$c.txt ] || continue
    cp $O/reads_kernels_timed_$c.txt $P/r06_reads_kernels_timed_$c.txt
    cp $(ls $O/reads_trace_$c/*/*_kernel_stats.csv | head -1) $P/r06_reads_kernel_stats_$c.csv
    cp $O/reads_$c.json $P/r06_reads_$c.json
    [ -f $O/reads_one_kernel_$c.json ] && cp $O/reads_one_kernel_$c.json $P/r06_reads_one_kernel_$c.json
  done
  for f in call_sq_counters_30x chain_sq_counters_30x reads_chain_sq_counters_30x; do [ -f $O/$f.txt ] && cp $O/$f.txt $P/r06_$f.txt; done
  [ -f $O/accsum_sq_counters_30x.txt ] && cp $O/accsum_sq_counters_30x.txt $P/r06_acc_sq_counters_30x.txt
  for f in bench bench_cfg4_10Mb_200x bench_cfg1_1Mb_10x cfg3_rank0of8 cfg5_rank0of8; do
    [ -f $O/$f.json ] && cp $O/$f.json $P/r06_${f#bench_}.json
  done
  [ -f $O/small_blocks.txt ] && { grep -v amdgpu.ids $O/small_blocks.txt > $P/r06_small_blocks.txt; cp $O/small_blocks.json $P/r06_small_blocks.json; }
fi
sed -i "s#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#; s#/root/repo/gpurun_out/#gpurun_out/#" $P/r06_*_timed.txt 2>/dev/null || true
python3 tools/make_resources.py > /dev/null 2>&1 || true
echo collected
