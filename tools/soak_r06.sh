#!/bin/bash
# One soak of each fuzz tool against the CPU oracle (the encoder's: against the host encoder and py_bcf; the device BAM reader's: against the host
# reader and py_bam), the six side by side on the one GPU (GPU box).  usage: bash tools/soak_r06.sh <tag> <minutes> <seed>
TAG=$1; MIN=${2:-5.5}; SEED=${3:-600}
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
P=()
k=0
for t in chain block pipeline reads bcf bamdev; do
  timeout -k 10 900 python3 tools/fuzz_$t.py --minutes $MIN --seed $((SEED + k)) > $O/fuzz_$t.txt 2>&1 & P+=($!)
  k=$((k + 1))
done
alive() { for p in "${P[@]}"; do kill -0 $p 2>/dev/null && return 0; done; return 1; }
while alive; do sleep 45; tail -qn 1 $O/fuzz_*.txt | cut -c1-100; done
rc=0
for p in "${P[@]}"; do wait $p || rc=1; done
tail -n 2 $O/fuzz_*.txt
exit $rc
