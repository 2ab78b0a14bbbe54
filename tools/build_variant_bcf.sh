#!/bin/bash
# A variant of the BCF encoder's kernels next to the main library: tools/build_variant_bcf.sh <name> <extra hipcc flags, e.g. -DBCF_WPE_SITES=5 -DBCF_IMG_SITES=6144u>
# -> bs_call_amd/lib/variants/lib_<name>.so (select with BSCALL_AMD_LIB=...); every other object is the main library's
set -e
NAME=$1; shift
D=bs_call_amd/lib/variants; L=bs_call_amd/lib
mkdir -p $D
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wno-unused-function -Wno-inline-asm "$@" -c bs_call_amd/csrc/bcfdev.hip -o $D/bcfdev_$NAME.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: bsc_bcf" | grep -E "Name|VGPRs:|Scratch|Occupancy|LDS" | sed "s/^.*remark: */[$NAME] /"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$NAME.so $L/kernels.o $L/fused.o $L/accumulate.o $L/sort.o $L/vcfcore.o $L/sitestats.o $L/compact.o $L/probe.o $L/prepdev.o $D/bcfdev_$NAME.o $L/bamdev.o $L/bscall_api.o $L/bamstream.o $L/inflate_fast.o $L/synth_reads.o $L/vcf_format.o $L/dbsnp.o $L/prep.o $L/report.o $L/bcf.o $L/bamio.o $L/refseq.o -lm -lz -lpthread
ls -la $D/lib_$NAME.so
