#!/bin/bash
# the glue's protocols end to end (integration/demo_block): the gt_vcf form (amd_overlap_protocol.h) and the bytes form (amd_bcf_protocol.h,
# INTEGRATION.md 2b) against the mock threads; mock printer -1 = counts only; BSC_DEMO_MPROF_JOBS=4.  usage: tools/r06_glue.sh > out.txt
export BSC_DEMO_MPROF_JOBS=4
echo "# tools/r06_glue.sh (one box): integration/demo_block; gt_vcf form = 'glue protocol end to end', bytes form = 'bytes form end to end'"
for cfg in "10000 30 1200" "100000 30 120" "1000000 30 12"; do
  for ns in -1 0; do
    echo "== demo_block $cfg, mock printer $ns ns per position"
    BSC_DEMO_PRINT_NS=$ns bs_call_amd/lib/demo_block $cfg 2>&1 | grep -v "^chrS\|^block\|^statistics" || exit 1
  done
done
