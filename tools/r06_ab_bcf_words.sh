#!/bin/bash
# A/B on one box, interleaved: the BCF write kernel of the library (fields composed as words, the chain's byte per position as gate and
# length, BCF_WPE_SITES waves a SIMD) against variants built next to it (bs_call_amd/lib/variants/lib_<name>.so: tools/build_variant_bcf.sh;
# lib_bcf_head.so = the byte-by-byte form of the commit before: `git show 53f79b9^:bs_call_amd/csrc/bcfdev.hip` compiled and linked as
# build_variant_bcf.sh does); the streams' checksums must agree.
# usage: bash tools/r06_ab_bcf_words.sh <tag> [variant names ...]
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
VARS="$@"
mkdir -p $O
cd $GRAFT_REPO_ROOT
for k in 1 2 3; do
  timeout -k 10 200 python3 tools/bench_sites_bcf.py --steps 8 > $O/main_$k.json
  BSC_BCF_NO_GATE=1 timeout -k 10 200 python3 tools/bench_sites_bcf.py --steps 8 > $O/main-nogate_$k.json
  for v in $VARS; do
    BSCALL_AMD_LIB=$GRAFT_REPO_ROOT/bs_call_amd/lib/variants/lib_$v.so timeout -k 10 200 python3 tools/bench_sites_bcf.py --steps 8 > $O/${v}_$k.json
  done
done
python3 - $O main main-nogate $VARS <<'PY' | tee $O/ab_bcf_words.txt
import json, sys, glob
o = sys.argv[1]
rows = {v: [json.load(open(f)) for f in sorted(glob.glob("%s/%s_[0-9].json" % (o, v)))] for v in sys.argv[2:]}
allr = [x for v in rows.values() for x in v]
assert len({x["out_sum"] for x in allr}) == 1 and len({x["bcf_bytes"] for x in allr}) == 1, "streams differ"
print("# BCF per-position form, 50 M positions at 30x, stage ms (avg of 8 launches; three runs each, interleaved on one box); same stream checksum %d (%d bytes)" % (allr[0]["out_sum"], allr[0]["bcf_bytes"]))
for v, r in rows.items():
    print("%-14s %s" % (v, " / ".join("%.3f" % x["stage_ms_avg"] for x in r)))
PY
