# Builds libbscall_amd.so (gfx950 kernels + C host code) in-tree, and the CPU oracle used by the tests.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CC ?= gcc
ARCH ?= gfx950
CSRC = bs_call_amd/csrc
LIBDIR = bs_call_amd/lib

# -ffp-contract=off: the reference's arithmetic has no fused operations (x86-64 baseline build); every
# fused multiply-add in the kernels is an explicit __builtin_fma in bsmath.h.
HIPFLAGS = -O3 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wall -Wno-unused-function -Wno-inline-asm
CFLAGS = -O2 -fPIC -Wall -ffp-contract=off -std=gnu11 -I$(ROCM)/include -D__HIP_PLATFORM_AMD__

all: $(LIBDIR)/libbscall_amd.so oracle demo

$(LIBDIR)/kernels.o: $(CSRC)/kernels.hip $(CSRC)/callmath.h $(CSRC)/call_body.inc $(CSRC)/call_summary.inc $(CSRC)/bsmath.h $(CSRC)/bsmath_tables.h $(CSRC)/devtables.h $(CSRC)/synth.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/bscall_api.o: $(CSRC)/bscall_api.c include/bscall_amd.h $(CSRC)/bsmath_tables.h $(CSRC)/devtables.h $(CSRC)/synth.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/accumulate.o: $(CSRC)/accumulate.hip $(CSRC)/accdev.h $(CSRC)/devtables.h $(CSRC)/call_summary.inc
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# rocPRIM's device radix sort (ROCm's primitive library): orders a block's templates by leftmost position
$(LIBDIR)/sort.o: $(CSRC)/sort.hip
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Wno-deprecated-declarations -c $< -o $@

$(LIBDIR)/vcfcore.o: $(CSRC)/vcfcore.hip $(CSRC)/devtables.h $(CSRC)/bsmath.h $(CSRC)/bsmath_tables.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/sitestats.o: $(CSRC)/sitestats.hip $(CSRC)/sitestats_dev.h $(CSRC)/devtables.h $(CSRC)/bsmath.h $(CSRC)/bsmath_tables.h include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/fused.o: $(CSRC)/fused.hip $(CSRC)/accdev.h $(CSRC)/callmath.h $(CSRC)/call_body.inc $(CSRC)/call_summary.inc $(CSRC)/sitestats_dev.h $(CSRC)/devtables.h $(CSRC)/bsmath.h $(CSRC)/bsmath_tables.h include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Wno-unused-variable -c $< -o $@

$(LIBDIR)/prepdev.o: $(CSRC)/prepdev.hip include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/bcfdev.o: $(CSRC)/bcfdev.hip include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/probe.o: $(CSRC)/probe.hip
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/compact.o: $(CSRC)/compact.hip include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/vcf_format.o: $(CSRC)/vcf_format.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/dbsnp.o: $(CSRC)/dbsnp.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/prep.o: $(CSRC)/prep.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/report.o: $(CSRC)/report.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/bcf.o: $(CSRC)/bcf.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/bamio.o: $(CSRC)/bamio.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/bamstream.o: $(CSRC)/bamstream.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/inflate_fast.o: $(CSRC)/inflate_fast.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -O3 -c $< -o $@

$(LIBDIR)/bamdev.o: $(CSRC)/bamdev.hip $(CSRC)/bamdev_core.h include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Wno-deprecated-declarations -c $< -o $@

$(LIBDIR)/refseq.o: $(CSRC)/refseq.c include/bscall_amd.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/synth_reads.o: $(CSRC)/synth_reads.c include/bscall_amd.h $(CSRC)/synth.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/libbscall_amd.so: $(LIBDIR)/kernels.o $(LIBDIR)/fused.o $(LIBDIR)/accumulate.o $(LIBDIR)/sort.o $(LIBDIR)/vcfcore.o $(LIBDIR)/sitestats.o $(LIBDIR)/compact.o $(LIBDIR)/probe.o $(LIBDIR)/prepdev.o $(LIBDIR)/bcfdev.o $(LIBDIR)/bamdev.o $(LIBDIR)/bscall_api.o $(LIBDIR)/bamstream.o $(LIBDIR)/inflate_fast.o $(LIBDIR)/synth_reads.o $(LIBDIR)/vcf_format.o $(LIBDIR)/dbsnp.o $(LIBDIR)/prep.o $(LIBDIR)/report.o $(LIBDIR)/bcf.o $(LIBDIR)/bamio.o $(LIBDIR)/refseq.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lm -lz -lpthread

oracle:
	$(MAKE) -C oracle liboracle.so

# The library with its HOST C files under AddressSanitizer + UndefinedBehaviorSanitizer (gcc's runtimes; the device objects
# are the ordinary ones): what tests/test_host_sanitizers.py and tools/fuzz_host_inputs.py run the readers and the host
# logic under, on the CPU.  Load it with LD_PRELOAD=<libasan.so>:<libubsan.so> BSCALL_AMD_LIB=$(LIBDIR)/san/libbscall_amd_san.so.
SANFLAGS = -O1 -g -fPIC -Wall -ffp-contract=off -std=gnu11 -I$(ROCM)/include -D__HIP_PLATFORM_AMD__ -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
HOST_C = bscall_api synth_reads vcf_format dbsnp prep report bcf bamio bamstream inflate_fast refseq
san: $(LIBDIR)/libbscall_amd.so
	@mkdir -p $(LIBDIR)/san
	for f in $(HOST_C); do $(CC) $(SANFLAGS) -c $(CSRC)/$$f.c -o $(LIBDIR)/san/$$f.o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(LIBDIR)/san/libbscall_amd_san.so $(LIBDIR)/kernels.o $(LIBDIR)/fused.o $(LIBDIR)/accumulate.o $(LIBDIR)/sort.o $(LIBDIR)/vcfcore.o $(LIBDIR)/sitestats.o $(LIBDIR)/compact.o $(LIBDIR)/probe.o $(LIBDIR)/prepdev.o $(LIBDIR)/bcfdev.o $(LIBDIR)/bamdev.o $(addprefix $(LIBDIR)/san/,$(addsuffix .o,$(HOST_C))) -L$(dir $(shell $(CC) -print-file-name=libasan.so)) -lasan -lubsan -lm -lz -lpthread

# a plain-C host program against the C ABI: gcc only, links the shared library like bs_call would
demo: $(LIBDIR)/demo_block $(LIBDIR)/bam2bcf
$(LIBDIR)/demo_block: integration/demo_block.c integration/mock_work.h integration/amd_overlap_protocol.h integration/amd_bcf_protocol.h include/bscall_amd.h $(LIBDIR)/libbscall_amd.so
	$(CC) -O2 -Wall -std=gnu11 -Iinclude -Iintegration $< -o $@ -L$(LIBDIR) -lbscall_amd -lpthread -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib
# BAM + FASTA -> BCF + JSON report with nothing but the C ABI (the C twin of bs_call_amd/pipeline.py)
bam2bcf: $(LIBDIR)/bam2bcf
$(LIBDIR)/bam2bcf: integration/bam2bcf.c include/bscall_amd.h $(LIBDIR)/libbscall_amd.so
	$(CC) -O2 -Wall -std=gnu11 -Iinclude $< -o $@ -L$(LIBDIR) -lbscall_amd -lpthread -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib

# Compile check of the two replacement translation units against the reference's own headers.  Only in a
# container that has /root/reference; the reference's bs_call.h includes three htslib headers for POINTER TYPES only, so
# the check gives the compiler opaque typedefs for them (generated under /tmp, never committed).  It checks OUR files'
# syntax and types against the reference's declarations; it claims nothing about the oracle or about the reference.
REF ?= /root/reference
glue-check:
	@if [ -d $(REF)/include ]; then \
	  d=$$(mktemp -d /tmp/bsc_glue.XXXXXX); mkdir -p $$d/htslib; \
	  printf 'typedef struct bam_hdr_t bam_hdr_t; typedef struct htsFile htsFile; typedef struct hts_idx_t hts_idx_t; typedef struct hts_itr_t hts_itr_t; typedef struct bam1_t bam1_t;\n#define FT_UNKN 0\n#define FT_GZ 1\n#define FT_VCF 2\n#define FT_VCF_GZ 3\n#define FT_BCF 4\n#define FT_BCF_GZ 5\n' > $$d/htslib/sam.h; \
	  printf 'typedef struct bcf_hdr_t bcf_hdr_t; typedef struct bcf1_t bcf1_t;\n' > $$d/htslib/vcf.h; \
	  printf 'typedef struct faidx_t faidx_t;\n' > $$d/htslib/faidx.h; \
	  for f in integration/call_genotypes_amd.c integration/call_genotypes_amd_overlap.c integration/call_genotypes_amd_bcf.c; do \
	    $(CC) -std=gnu11 -Wall -fsyntax-only -D__LINUX__ -DAMD_GLUE_CHECK -I$$d -I$(REF)/include -I$(REF)/gt/include -Iinclude $$f && echo "glue-check: $$f ok" || exit 1; \
	  done; rm -rf $$d; \
	else echo "glue-check: $(REF) not present, skipped"; fi

asm: $(CSRC)/kernels.hip
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage $< -o $(LIBDIR)/kernels.s

clean:
	rm -rf $(LIBDIR)/san; rm -f $(LIBDIR)/*.o $(LIBDIR)/*.so $(LIBDIR)/*.s $(LIBDIR)/demo_block $(LIBDIR)/bam2bcf
	$(MAKE) -C oracle clean

.PHONY: all oracle demo asm clean glue-check san
