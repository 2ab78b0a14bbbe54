#!/bin/bash
# The kernels of a COMMITTED state as a library variant next to the working tree's (A/B of the working tree against it):
# tools/build_variant_head.sh <name> [<commit>]  ->  bs_call_amd/lib/variants/lib_<name>.so  (kernels from the commit, host objects from lib/)
set -e
NAME=$1; REV=${2:-HEAD}
T=$(mktemp -d /tmp/bsc_var.XXXXXX)
git archive $REV bs_call_amd/csrc include | tar -x -C $T
D=bs_call_amd/lib/variants; L=bs_call_amd/lib
mkdir -p $D
F="-O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wno-unused-function -Wno-unused-variable -Wno-inline-asm"
for f in fused kernels accumulate; do /opt/rocm/bin/hipcc $F -c $T/bs_call_amd/csrc/$f.hip -o $D/${f}_$NAME.o 2>/dev/null; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$NAME.so $D/fused_$NAME.o $D/kernels_$NAME.o $D/accumulate_$NAME.o $L/sort.o $L/vcfcore.o $L/sitestats.o $L/compact.o $L/probe.o $L/prepdev.o $L/bcfdev.o $L/bscall_api.o $L/synth_reads.o $L/vcf_format.o $L/dbsnp.o $L/prep.o $L/report.o $L/bcf.o $L/bamio.o $L/refseq.o -lm -lz -lpthread 2>&1 | grep -v warning || true
rm -rf $T
ls -la $D/lib_$NAME.so
