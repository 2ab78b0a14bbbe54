#!/bin/bash
# After a kernel change (GPU box): the whole GPU suite, then the chain / reads / bench timings un-profiled and the VALU count of
# the summary-in chain kernel.  usage: bash tools/r04_check.sh <tag>   (SKIP_TESTS=1: timings only)
set -e
TAG=$1
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
[ -n "$SKIP_TESTS" ] || { timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }; tail -2 $O/pytest.txt; }
timeout -k 10 300 python3 tools/bench_chain.py --no-unfused --steps 20 --warm 8 > $O/chain.json 2> $O/chain.err || { tail -5 $O/chain.err; exit 1; }
timeout -k 10 300 python3 tools/bench_reads.py --steps 10 --warm 10 --no-check > $O/reads.json 2> $O/reads.err || { tail -5 $O/reads.err; exit 1; }
timeout -k 10 300 python3 tools/bench_reads.py --one-kernel --steps 10 --warm 10 --no-check > $O/reads1.json 2>> $O/reads.err || { tail -5 $O/reads.err; exit 1; }
timeout -k 10 400 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-chain > $O/bench_nochain.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<PY
import json
ch = json.loads(open("$O/chain.json").read().strip().splitlines()[-1])
r = json.loads(open("$O/reads.json").read().strip().splitlines()[-1])
r1 = json.loads(open("$O/reads1.json").read().strip().splitlines()[-1])
b = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("chain", {k: ch[k] for k in ch if "ms" in k})
print("reads two-kernel %.3f ms (min %.3f)  one-kernel %.3f ms  accumulate %.3f ms" % (r["reads_chain"]["device_ms_avg"], r["reads_chain"]["device_ms_min"], r1["reads_chain"]["device_ms_avg"], r["accumulate"]["device_ms_avg"]))
b2 = json.loads(open("$O/bench_nochain.json").read().strip().splitlines()[-1])
print("bench --no-chain: acc %.3f reads %.3f" % (b2["roofline_accumulate"]["stage_ms_avg"], b2["roofline_reads"]["stage_ms_avg"]))
print("bench value %.3f G  call %.3f ms  chain %.3f  acc %.3f  reads %.3f" % (b["value"] / 1e9, b["roofline"]["kernel_ms_avg"], b["roofline_chain"]["kernel_ms_avg"], b["roofline_accumulate"]["stage_ms_avg"], b["roofline_reads"]["stage_ms_avg"]))
PY
bash tools/pmc_kernel.sh ${TAG}_rc "bsc_chain_kernel_t<true, false, false, true>" tools/bench_reads.py --steps 2 --no-check > $O/sq.txt 2>&1 || { tail $O/sq.txt; exit 1; }
grep -E 'SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES' $O/sq.txt
