#!/bin/bash
# Round 4 mid-round check (GPU box): the whole GPU suite, then SQ counters and FETCH / WRITE traffic of the reads-in chain and
# the stand-alone accumulate kernel at configs[1] size.  usage: bash tools/r04_mid.sh <tag>
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O/prof
[ -n "$SKIP_TESTS" ] || { timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }; tail -2 $O/pytest.txt; }
prof() { local out=$1; shift; (cd /tmp && rocprofv3 "$@" > $O/$out.stdout 2> $O/$out.err) || { tail -5 $O/$out.err; exit 1; }; }
bash tools/pmc_kernel.sh ${TAG}_rc30 "bsc_chain_kernel_t<true, true>" tools/bench_reads.py --steps 2 --no-check > $O/reads_chain_sq_counters_30x.txt 2>&1 || { tail $O/reads_chain_sq_counters_30x.txt; exit 1; }
cat $O/reads_chain_sq_counters_30x.txt
prof reads_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/reads_fetch -- python3 $ROOT/tools/bench_reads.py --steps 2 --no-check
prof reads_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/reads_write -- python3 $ROOT/tools/bench_reads.py --steps 2 --no-check
python3 - <<PY
import csv, glob
for sub, name in (("reads_fetch", "FETCH_SIZE"), ("reads_write", "WRITE_SIZE")):
    for kern in ("bsc_chain_kernel_t<true, true>", "bsc_accumulate_kernel"):
        agg = {}
        for f in glob.glob("$O/prof/%s/**/*counter_collection.csv" % sub, recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == name and kern in r.get("Kernel_Name", ""):
                    agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        v = list(agg.values())
        print("%-34s %-10s %.4g KiB per dispatch (%d dispatches)" % (kern, name, sum(v) / max(len(v), 1), len(v)))
PY
