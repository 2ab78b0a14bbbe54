#!/bin/bash
# A sharded run of integration/bam2bcf over one BAM file: one rank per GPU (HIP_VISIBLE_DEVICES), no collective between them, then the merge.
# usage: tools/bam2bcf_sharded.sh N in.bam ref.fa out.bcf report.json [sample]     (BAM2BCF_ONE_GPU=1: every rank on GPU 0 — a rehearsal)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
pids=()
for ((r = 0; r < N; r++)); do
  if [ -n "$BAM2BCF_ONE_GPU" ]; then dev=0; else dev=$r; fi
  HIP_VISIBLE_DEVICES=$dev $ROOT/bs_call_amd/lib/bam2bcf --rank $r --world $N "$@" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "a rank failed" >&2; exit 1; }; done
$ROOT/bs_call_amd/lib/bam2bcf --merge $N "$@"
