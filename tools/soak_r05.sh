#!/bin/bash
# One soak of each fuzz tool against the CPU oracle (the encoder's: against the host encoder and py_bcf), the five side by side on the one GPU (GPU box).
# usage: bash tools/soak_r05.sh <tag> <minutes> <seed>
TAG=$1; MIN=${2:-5.5}; SEED=${3:-500}
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
timeout -k 10 900 python3 tools/fuzz_chain.py --minutes $MIN --seed $SEED > $O/fuzz_chain.txt 2>&1 & P1=$!
timeout -k 10 900 python3 tools/fuzz_block.py --minutes $MIN --seed $((SEED + 1)) > $O/fuzz_block.txt 2>&1 & P2=$!
timeout -k 10 900 python3 tools/fuzz_pipeline.py --minutes $MIN --seed $((SEED + 2)) > $O/fuzz_pipeline.txt 2>&1 & P3=$!
timeout -k 10 900 python3 tools/fuzz_reads.py --minutes $MIN --seed $((SEED + 3)) > $O/fuzz_reads.txt 2>&1 & P4=$!
timeout -k 10 900 python3 tools/fuzz_bcf.py --minutes $MIN --seed $((SEED + 4)) > $O/fuzz_bcf.txt 2>&1 & P5=$!
rc=0
while kill -0 $P5 2>/dev/null || kill -0 $P1 2>/dev/null || kill -0 $P2 2>/dev/null || kill -0 $P3 2>/dev/null || kill -0 $P4 2>/dev/null; do sleep 45; tail -qn 1 $O/fuzz_*.txt | cut -c1-100; done
for p in $P1 $P2 $P3 $P4 $P5; do wait $p || rc=1; done
tail -n 2 $O/fuzz_*.txt
exit $rc
