#!/bin/bash
# The round's closing run on one box: the whole GPU suite, smoke(), a soak of tools/fuzz_block.py alone (fresh contexts: the create race's
# regression), the packed-form encoder's traffic passes (tools/pmc_bcf.sh), then the plain bench line.  usage: bash tools/r06_final.sh <tag>
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -30 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log | cut -c1-300
timeout -k 10 400 python3 tools/fuzz_block.py --minutes 4 --seed 31000 > $O/fuzz_block.txt 2>&1 || { tail -5 $O/fuzz_block.txt; exit 1; }
tail -1 $O/fuzz_block.txt
# the per-position form's traffic passes (as part c of tools/prof_r06.sh) and the packed form's: traffic.json entries bcf_sites / bcf on these sources
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/sites_$c -- python3 $ROOT/tools/bench_sites_bcf.py --steps 2 > $O/sites_$c.out 2> $O/sites_$c.err) || { tail -5 $O/sites_$c.err; exit 1; }
done
python3 tools/make_r06_json.py $O | tee $O/r06_json.txt | tail -4
bash tools/pmc_bcf.sh $TAG/pmc_bcf > $O/pmc_bcf.log 2>&1 || { tail -5 $O/pmc_bcf.log; exit 1; }
tail -4 $O/pmc_bcf.log
cp profiles/traffic.json $O/traffic.json
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
echo final done
