#!/bin/bash
# Build a variant of the fused chain kernel next to the main library: tools/build_variant_fused.sh <name> <extra hipcc flags...>
# -> bs_call_amd/lib/variants/lib_<name>.so  (select with BSCALL_AMD_LIB=...)
set -e
NAME=$1; shift
D=bs_call_amd/lib/variants
mkdir -p $D
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wno-unused-function -Wno-unused-variable "$@" -c bs_call_amd/csrc/fused.hip -o $D/f_$NAME.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "bsc_chain_kernel_tILb1" | grep -E "VGPRs:|Scratch|Occupancy|SGPRs Spill|LDS" | sed "s/^.*remark: */[$NAME] /"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wno-unused-function "$@" -c bs_call_amd/csrc/accumulate.hip -o $D/a_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$NAME.so $D/f_$NAME.o bs_call_amd/lib/kernels.o $D/a_$NAME.o bs_call_amd/lib/sort.o bs_call_amd/lib/vcfcore.o bs_call_amd/lib/sitestats.o bs_call_amd/lib/compact.o bs_call_amd/lib/probe.o bs_call_amd/lib/prepdev.o bs_call_amd/lib/bcfdev.o bs_call_amd/lib/bamdev.o bs_call_amd/lib/bscall_api.o bs_call_amd/lib/bamstream.o bs_call_amd/lib/inflate_fast.o bs_call_amd/lib/synth_reads.o bs_call_amd/lib/vcf_format.o bs_call_amd/lib/dbsnp.o bs_call_amd/lib/prep.o bs_call_amd/lib/report.o bs_call_amd/lib/bcf.o bs_call_amd/lib/bamio.o bs_call_amd/lib/refseq.o -lm -lz -lpthread
