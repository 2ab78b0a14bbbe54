#!/bin/bash
# Build a kernel variant next to the main library: tools/build_variant.sh <name> <extra hipcc flags...>
# -> bs_call_amd/lib/variants/lib_<name>.so  (select with BSCALL_AMD_LIB=...)
set -e
NAME=$1; shift
D=bs_call_amd/lib/variants
mkdir -p $D
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wno-unused-function "$@" -c bs_call_amd/csrc/kernels.hip -o $D/k_$NAME.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "bsc_call_kernel_tILb1" | grep -E "VGPRs:|Scratch|Occupancy" | sed "s/^.*remark: */[$NAME] /"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$NAME.so $D/k_$NAME.o bs_call_amd/lib/fused.o bs_call_amd/lib/accumulate.o bs_call_amd/lib/sort.o bs_call_amd/lib/vcfcore.o bs_call_amd/lib/sitestats.o bs_call_amd/lib/compact.o bs_call_amd/lib/probe.o bs_call_amd/lib/prepdev.o bs_call_amd/lib/bcfdev.o bs_call_amd/lib/bscall_api.o bs_call_amd/lib/synth_reads.o bs_call_amd/lib/vcf_format.o bs_call_amd/lib/dbsnp.o bs_call_amd/lib/prep.o bs_call_amd/lib/report.o bs_call_amd/lib/bcf.o bs_call_amd/lib/bamio.o bs_call_amd/lib/refseq.o -lm -lz -lpthread

# fused-chain variants: tools/build_variant.sh --fused <name> <flags>   (wrapper below)
