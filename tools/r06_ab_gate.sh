#!/bin/bash
# A/B on one box: the BCF write kernel with the chain's byte per position as its gate (default) against the form that finds a position's
# flag in its record (BSC_BCF_NO_GATE), interleaved; the streams' checksums must agree.  usage: bash tools/r06_ab_gate.sh <tag>
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
cd $GRAFT_REPO_ROOT
for k in 1 2 3; do
  timeout -k 10 200 python3 tools/bench_sites_bcf.py --steps 8 > $O/gate_$k.json
  BSC_BCF_NO_GATE=1 timeout -k 10 200 python3 tools/bench_sites_bcf.py --steps 8 > $O/nogate_$k.json
done
python3 - $O <<'PY'
import json, sys, glob
o = sys.argv[1]
g = [json.load(open(f)) for f in sorted(glob.glob(o + "/gate_*.json"))]
n = [json.load(open(f)) for f in sorted(glob.glob(o + "/nogate_*.json"))]
assert len({x["out_sum"] for x in g + n}) == 1 and len({x["bcf_bytes"] for x in g + n}) == 1, "streams differ"
line = "BCF per-position form, 50 M positions at 30x, stage ms (avg of 8, three runs each, interleaved): gate byte %s   flag in the record %s   same stream checksum %d (%d bytes)" % (
    " / ".join("%.3f" % x["stage_ms_avg"] for x in g), " / ".join("%.3f" % x["stage_ms_avg"] for x in n), g[0]["out_sum"], g[0]["bcf_bytes"])
print(line)
open(o + "/ab_bcf_gate.txt", "w").write(line + "\n")
PY
