/*
 * bam2bcf.c — plain-C host program against libbscall_amd.so (gcc only): the reference's data path from a coordinate-sorted
 * BAM file and a FASTA reference to an uncompressed BCF stream and the JSON report, block by block, with nothing but the
 * C ABI of include/bscall_amd.h — what bs_call's four threads do between sam_read1() and bcf_write():
 *
 *   reader thread     bsc_bamdev_next_block         read_input / get_next_align_details ON THE DEVICE (round 6): the host inflates the BGZF
 *                                                   blocks (BAM2BCF_THREADS helpers, default: one per core), the records become blocks of
 *                                                   raw templates in HBM (csrc/bamdev.hip).  BAM2BCF_HOST_READER: bsc_bam_next_block, the
 *                                                   host reader of rounds 2-5 (csrc/bamio.c) — same bytes
 *   process thread    bsc_block_reference           get_sequence_string
 *   process + calc    bsc_block_bcf_raw             process_template_vector + meth_profile (on the device since round 5:
 *     + print                                       bsc_prepare_templates_device), call_genotypes_ML, _print_vcf_entry WITH the
 *                                                   encoding (the bcf_enc_* calls and bcf_write's fixed fields, csrc/bcfdev.hip), the
 *                                                   statistics: the block's BCF bytes come back
 *                                                   (BAM2BCF_HOST_BCF: bsc_block_records_raw, then bsc_bcf_block on this thread;
 *                                                   BAM2BCF_HOST_PREP: also the pre-processing on this thread,
 *                                                   bsc_prepare_templates_profile + bsc_block_records, as in round 4)
 *   output            fwrite                        bcf_write's write
 *   at the end        bsc_report_json               output_stats
 *
 * Not a replacement of the bs_call executable (no option parsing, regions, contig lists, dbSNP, compression): a worked
 * example of the calls in order, and the C twin of bs_call_amd/pipeline.py — tests/test_gpu_pipeline.py checks that both
 * write the same bytes.
 *
 *   make bam2bcf && bs_call_amd/lib/bam2bcf in.bam ref.fa out.bcf report.json [sample]
 *
 * A sharded run over ONE file (SURVEY.md 8e on real input; the reference's unit of parallelism is a process per contig set, README.md): rank r of n
 *   bam2bcf --rank r --world n in.bam ref.fa out.bcf report.json [sample]
 * takes its share of the contigs (whole contigs, longest first to the least loaded rank: bs_call_amd/shard.py's assignment), reads the
 * stretches of the file that hold them (bsc_bamdev_open_contigs: no index file), writes every contig's records to out.bcf.shardNNNNN and what
 * the report sums over to out.bcf.rankNNNNN.sums; no rank talks to another.  When all have finished,
 *   bam2bcf --merge n in.bam ref.fa out.bcf report.json [sample]
 * writes the header, the shards behind it in the header's contig order, and the report from the ranks' sums — the bytes of the single run
 * (tests/test_gpu_shard_bam.py).  tools/bam2bcf_sharded.sh starts the ranks, one per GPU, and merges.
 * The header's date lines are left out (the reference's --benchmark-mode) so that the output is reproducible.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <bscall_amd.h>

#define CHECK(call)                                                          \
  do {                                                                       \
    long rc_ = (long)(call);                                                 \
    if (rc_ < 0) {                                                           \
      fprintf(stderr, "%s failed (%ld): %s\n", #call, rc_, bsc_last_error()); \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void *xrealloc(void *p, size_t n) {
  void *q = realloc(p, n ? n : 1);
  if (!q) {
    fprintf(stderr, "out of memory\n");
    exit(1);
  }
  return q;
}

static void *pinned(size_t n) { /* page-locked: the copy-out is a true DMA, queued behind the kernels */
  void *p = bsc_alloc_host(n);
  if (!p) {
    fprintf(stderr, "out of page-locked memory (%zu bytes)\n", n);
    exit(1);
  }
  return p;
}

/* The output thread (bcf_write's write).  A block's stream stays on the device (bsc_block_bcf_rawdev_keep) and is HANDED to this thread
 * (bsc_bcf_stream_detach): it reads it out in pieces of 32 MB — two page-locked buffers, the next piece's copy in flight on a stream of
 * its own while the last is written — and writes each at its offset of its file, then gives the buffer back (bsc_detached_free: the next
 * blocks take theirs from what came back).  The main thread meanwhile pulls the next block out of the reader and calls it: on a file of
 * several contigs the inflate (the reader's bound) and the write (this thread's: ~10 GB/s into the page cache) overlap instead of taking
 * turns — round 6's first form read the pieces on the main thread, which therefore waited for this one with the reader's slabs full.
 * ONE writer: three of them writing pieces of one file in parallel were 2.5 x slower (0.70 s against 0.28 for the 2.9 GB of a contig-sized
 * block: writes to one inode take turns anyway).  Page-locking a host buffer for a whole stream would cost more than the calling. */
#define PIECE ((size_t)32 << 20)
#define N_JOB 3
typedef struct {
  void *d;        /* the stream on the device; NULL: only close_fd */
  uint64_t n, at; /* its length, where it goes in the file */
  int fd, close_fd;
} out_job;
typedef struct {
  bsc_context *ctx;
  uint8_t *buf[2];
  out_job job[N_JOB];
  int head, count, quit, failed, started;
  double t_wait, t_write, t_idle; /* the output thread's own account (BAM2BCF_TIMING) */
  pthread_mutex_t mu;
  pthread_cond_t cv;
  pthread_t th;
} out_writer;
static double now(void);
static void *writer_main(void *a) {
  out_writer *w = a;
  for (;;) {
    double t0 = now(), t1;
    pthread_mutex_lock(&w->mu);
    while (!w->count && !w->quit) pthread_cond_wait(&w->cv, &w->mu);
    w->t_idle += (t1 = now()) - t0;
    if (!w->count) {
      pthread_mutex_unlock(&w->mu);
      return NULL;
    }
    const out_job j = w->job[w->head];
    pthread_mutex_unlock(&w->mu);
    int bad = 0;
    if (j.d) {
      uint64_t off = 0;
      int k = 0;
      size_t take = j.n < PIECE ? (size_t)j.n : PIECE;
      if (take && bsc_detached_read(w->ctx, j.d, 0, take, w->buf[0]) < 0) bad = 1;
      while (off < j.n && !bad) {
        t0 = now();
        if (bsc_detached_wait(w->ctx) < 0) bad = 1; /* piece k is here */
        w->t_wait += (t1 = now()) - t0;
        const uint64_t next = off + take;
        const size_t take2 = j.n - next < PIECE ? (size_t)(j.n - next) : PIECE;
        if (!bad && take2 && bsc_detached_read(w->ctx, j.d, next, take2, w->buf[k ^ 1]) < 0) bad = 1; /* the next one: while this one is written */
        size_t done = 0;
        while (done < take && !bad) {
          const ssize_t r = pwrite(j.fd, w->buf[k] + done, take - done, (off_t)(j.at + off + done));
          if (r <= 0) bad = 1;
          else done += (size_t)r;
        }
        w->t_write += now() - t1;
        off = next;
        take = take2;
        k ^= 1;
      }
      if (bsc_detached_free(w->ctx, j.d) < 0) bad = 1;
    }
    if (j.close_fd && j.fd >= 0) close(j.fd);
    pthread_mutex_lock(&w->mu);
    if (bad) w->failed = 1;
    w->head = (w->head + 1) % N_JOB;
    w->count--;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
  }
}
static void writer_push(out_writer *w, out_job j) {
  pthread_mutex_lock(&w->mu);
  while (w->count == N_JOB) pthread_cond_wait(&w->cv, &w->mu);
  w->job[(w->head + w->count) % N_JOB] = j;
  w->count++;
  pthread_cond_broadcast(&w->cv);
  pthread_mutex_unlock(&w->mu);
}

/* The next contig's reference in the background (load_sequence + the GC bins): a thread of its own reads the FASTA while the reader's helpers
 * inflate.  One request in flight. */
typedef struct {
  const char *fasta, *name;
  uint64_t want;
  uint8_t *codes, *gc;
  uint64_t codes_len, n_bins;
  uint32_t gc_start;
  int rc, tid;
  char err[256];
  pthread_t th;
  int running;
} ref_job;
static void *ref_main(void *a) {
  ref_job *j = a;
  j->codes = malloc(j->want ? j->want : 1);
  j->gc = NULL;
  j->rc = j->codes ? bsc_fasta_contig(j->fasta, j->name, j->codes, j->want, &j->codes_len) : -1;
  if (j->rc >= 0) {
    j->gc = malloc((size_t)(j->codes_len / 100 + 1));
    j->rc = j->gc ? bsc_gc_bins(j->codes, j->codes_len, &j->gc_start, j->gc, j->codes_len / 100 + 1, &j->n_bins) : -1;
  }
  if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", bsc_last_error());
  return NULL;
}
static void ref_start(ref_job *j, const char *fasta, const char *name, uint64_t len, int tid) {
  memset(j, 0, sizeof *j);
  j->fasta = fasta;
  j->name = name;
  j->want = len;
  j->tid = tid;
  j->running = pthread_create(&j->th, NULL, ref_main, j) == 0;
  if (!j->running) ref_main(j);
}

/* the header print_vcf_header assembles in --benchmark-mode (src/print_vcf.c:621-745) */
static void write_header(FILE *f, int n_refs, const char *const *names, const uint32_t *lens, const char *sample) {
  static const char *const defs[] = {
      "##INFO=<ID=CX,Number=1,Type=String,Description=\"5 base sequence context (from position -2 to +2 on the positive strand) determined from the reference\">",
      "##FILTER=<ID=fail,Description=\"No sample passed filters\">",
      "##FILTER=<ID=q20,Description=\"Genotype Quality below 20\">",
      "##FILTER=<ID=qd2,Description=\"Quality By Depth below 2\">",
      "##FILTER=<ID=fs60,Description=\"Fisher Strand above 60\">",
      "##FILTER=<ID=mq40,Description=\"RMS Mapping Quality below 40\">",
      "##FILTER=<ID=mac1,Description=\"Minor allele count <= 1\">",
      "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">",
      "##FORMAT=<ID=FT,Number=1,Type=String,Description=\"Sample Genotype Filter\">",
      "##FORMAT=<ID=GL,Number=G,Type=Float,Description=\"Genotype Likelihood\">",
      "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Phred scaled conditional genotype quality\">",
      "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read Depth (non converted reads only)\">",
      "##FORMAT=<ID=MQ,Number=1,Type=Integer,Description=\"RMS Mapping Quality\">",
      "##FORMAT=<ID=QD,Number=1,Type=Integer,Description=\"Quality By Depth (Variant quality / read depth (non-converted reads only))\">",
      "##FORMAT=<ID=MC8,Number=8,Type=Integer,Description=\"Base counts: non-informative for methylation (ACGT) followed by informative for methylation (ACGT)\">",
      "##FORMAT=<ID=AMQ,Number=.,Type=Integer,Description=\"Average base quailty for where MC8 base count non-zero\">",
      "##FORMAT=<ID=CS,Number=1,Type=String,Description=\"Strand of Cytosine relative to reference sequence (+/-/+-/NA)\">",
      "##FORMAT=<ID=CG,Number=1,Type=String,Description=\"CpG Status (from genotype calls: Y/N/H/?)\">",
      "##FORMAT=<ID=CX,Number=1,Type=String,Description=\"5 base sequence context (from position -2 to +2 on the positive strand) determined from genotype call\">",
      "##FORMAT=<ID=FS,Number=1,Type=Integer,Description=\"Phred scaled log p-value from Fishers exact test of strand bias\">"};
  size_t cap = 1 << 16, len = 0;
  char *t = xrealloc(NULL, cap);
#define ADD(...)                                                  \
  do {                                                            \
    for (;;) {                                                    \
      const int w_ = snprintf(t + len, cap - len, __VA_ARGS__);   \
      if ((size_t)w_ < cap - len) {                               \
        len += (size_t)w_;                                        \
        break;                                                    \
      }                                                           \
      cap *= 2;                                                   \
      t = xrealloc(t, cap);                                       \
    }                                                             \
  } while (0)
  ADD("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n");
  for (int i = 0; i < n_refs; i++) ADD("##contig=<ID=%s,length=%u>\n", names[i], lens[i]);
  for (size_t i = 0; i < sizeof defs / sizeof defs[0]; i++) ADD("%s\n", defs[i]);
  ADD("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n", sample);
#undef ADD
  const uint32_t l_text = (uint32_t)len + 1; /* the terminator is part of the BCF header text */
  fwrite("BCF\2\2", 1, 5, f);
  fwrite(&l_text, 4, 1, f); /* little-endian hosts only, like the rest of this example */
  fwrite(t, 1, len + 1, f);
  free(t);
}

/* ---- the sharded run ------------------------------------------------------------------------------------------------------------- */
#define PROF_CAP 4096
typedef struct { /* what a rank hands the merge: everything the report adds up */
  uint64_t magic, n_ref, n_blocks, n_records, malformed;
  uint64_t filter_cts[15], filter_bases[15], base_filter[5];
  uint64_t prof_used, prof[PROF_CAP][4];
  bsc_site_stats total;
  /* followed by: gc[BSC_COV_CAP * 101], then per contig {seen, totals[14]} x n_ref */
} rank_sums;
#define SUMS_MAGIC 0x3173635f6d756273ull

/* whole contigs to ranks, longest first, each to the least loaded rank (ties: the lower index) — bs_call_amd/shard.py assign_contigs */
static int *assign_contigs(const uint32_t *len, int n_ref, int world) {
  int *owner = calloc((size_t)n_ref + 1, sizeof *owner), *order = calloc((size_t)n_ref + 1, sizeof *order);
  uint64_t *load = calloc((size_t)world, sizeof *load);
  for (int i = 0; i < n_ref; i++) order[i] = i;
  for (int i = 1; i < n_ref; i++) { /* by (-length, index): an insertion sort keeps equal lengths in index order */
    const int v = order[i];
    int j = i;
    while (j > 0 && len[order[j - 1]] < len[v]) {
      order[j] = order[j - 1];
      j--;
    }
    order[j] = v;
  }
  for (int k = 0; k < n_ref; k++) {
    int best = 0;
    for (int r = 1; r < world; r++)
      if (load[r] < load[best]) best = r;
    owner[order[k]] = best;
    load[best] += len[order[k]];
  }
  free(order);
  free(load);
  return owner;
}

static void header_of(const char *bam_path, int *n_ref, const char ***names, uint32_t **lens, bsc_bamstream **keep) {
  bsc_bamstream *h = NULL;
  CHECK(bsc_bamstream_open_contigs(bam_path, 1, 0, 0, NULL, 0, &h)); /* no contig at all: the header alone */
  *n_ref = bsc_bamstream_n_refs(h);
  *names = calloc((size_t)*n_ref + 1, sizeof **names);
  *lens = calloc((size_t)*n_ref + 1, sizeof **lens);
  for (int i = 0; i < *n_ref; i++) {
    (*names)[i] = bsc_bamstream_ref_name(h, i);
    (*lens)[i] = bsc_bamstream_ref_len(h, i);
  }
  *keep = h; /* (the names live in it) */
}

static void write_header(FILE *f, int n_refs, const char *const *names, const uint32_t *lens, const char *sample);

static int merge_main(int world, char **argv, const char *sample) {
  int n_ref;
  const char **names;
  uint32_t *lens;
  bsc_bamstream *hs;
  header_of(argv[1], &n_ref, &names, &lens, &hs);
  FILE *out = fopen(argv[3], "wb");
  if (!out) {
    perror(argv[3]);
    return 1;
  }
  write_header(out, n_ref, names, lens, sample);
  char *path = malloc(strlen(argv[3]) + 64), *buf = malloc(1 << 24);
  for (int t = 0; t < n_ref; t++) { /* the contigs' shards in the header's order */
    sprintf(path, "%s.shard%05d", argv[3], t);
    FILE *f = fopen(path, "rb");
    if (!f) continue;
    size_t n;
    while ((n = fread(buf, 1, 1 << 24, f)) > 0) fwrite(buf, 1, n, out);
    fclose(f);
    remove(path);
  }
  fclose(out);
  /* the report: every number in it is a sum over positions / reads / templates, the read profile's length the longest any rank saw */
  rank_sums *acc = calloc(1, sizeof *acc), *one = malloc(sizeof *one);
  uint64_t *gc = calloc((size_t)BSC_COV_CAP * 101, 8), *gc1 = malloc((size_t)BSC_COV_CAP * 101 * 8);
  uint64_t *ct = calloc((size_t)n_ref * 15 + 1, 8), *ct1 = malloc(((size_t)n_ref * 15 + 1) * 8);
  for (int r = 0; r < world; r++) {
    sprintf(path, "%s.rank%05d.sums", argv[3], r);
    FILE *f = fopen(path, "rb");
    if (!f || fread(one, sizeof *one, 1, f) != 1 || one->magic != SUMS_MAGIC || one->n_ref != (uint64_t)n_ref ||
        fread(gc1, 8, (size_t)BSC_COV_CAP * 101, f) != (size_t)BSC_COV_CAP * 101 || fread(ct1, 8, (size_t)n_ref * 15, f) != (size_t)n_ref * 15) {
      fprintf(stderr, "bam2bcf --merge: %s is missing or not a rank's sums of this input\n", path);
      return 1;
    }
    fclose(f);
    remove(path);
    acc->n_blocks += one->n_blocks;
    acc->n_records += one->n_records;
    acc->malformed += one->malformed;
    for (int i = 0; i < 15; i++) acc->filter_cts[i] += one->filter_cts[i], acc->filter_bases[i] += one->filter_bases[i];
    for (int i = 0; i < 5; i++) acc->base_filter[i] += one->base_filter[i];
    if (one->prof_used > acc->prof_used) acc->prof_used = one->prof_used;
    for (int i = 0; i < PROF_CAP * 4; i++) (&acc->prof[0][0])[i] += (&one->prof[0][0])[i];
    { /* bsc_site_stats: the leading words are counters, the last 4 x 101 doubles (include/bscall_amd.h) */
      const size_t n_int = (sizeof(bsc_site_stats) - 404 * sizeof(double)) / 8;
      uint64_t *a = (uint64_t *)&acc->total;
      const uint64_t *b = (const uint64_t *)&one->total;
      for (size_t i = 0; i < n_int; i++) a[i] += b[i];
      double *ad = (double *)(a + n_int);
      const double *bd = (const double *)(b + n_int);
      for (int i = 0; i < 404; i++) ad[i] += bd[i];
    }
    for (size_t i = 0; i < (size_t)BSC_COV_CAP * 101; i++) gc[i] += gc1[i];
    for (size_t i = 0; i < (size_t)n_ref * 15; i++) ct[i] += ct1[i];
  }
  if (acc->malformed)
    fprintf(stderr, "bam2bcf: warning: %llu BAM records dropped, their CIGAR does not cover the sequence (damaged input?)\n", (unsigned long long)acc->malformed);
  bsc_report rep;
  memset(&rep, 0, sizeof rep);
  rep.under_conv = 0.01;
  rep.over_conv = 0.05;
  rep.mapq_thresh = 20;
  rep.min_qual = 20;
  rep.day = 1, rep.month = 1, rep.year = 2000;
  memcpy(rep.filter_cts, acc->filter_cts, sizeof rep.filter_cts);
  memcpy(rep.filter_bases, acc->filter_bases, sizeof rep.filter_bases);
  memcpy(rep.base_filter, acc->base_filter, sizeof rep.base_filter);
  rep.total = &acc->total;
  rep.gc = gc;
  rep.read_profile = &acc->prof[0][0];
  rep.n_read_profile = (uint32_t)acc->prof_used;
  bsc_contig_totals *listed = calloc((size_t)n_ref + 1, sizeof *listed);
  uint32_t nl = 0;
  for (int i = 0; i < n_ref; i++)
    if (ct[(size_t)i * 15]) {
      listed[nl].name = names[i];
      memcpy(listed[nl].snps, ct + (size_t)i * 15 + 1, 14 * 8);
      nl++;
    }
  rep.contigs = listed;
  rep.n_contigs = nl;
  const long need = bsc_report_json(&rep, NULL, 0);
  CHECK(need);
  char *text = xrealloc(NULL, (size_t)need + 1);
  bsc_report_json(&rep, text, (size_t)need + 1);
  FILE *fr = fopen(argv[4], "w");
  if (!fr) {
    perror(argv[4]);
    return 1;
  }
  fwrite(text, 1, (size_t)need, fr);
  fclose(fr);
  printf("%llu blocks, %llu records written\n", (unsigned long long)acc->n_blocks, (unsigned long long)acc->n_records);
  bsc_bamstream_close(hs);
  return 0;
}

int main(int argc, char **argv) {
  int rank = -1, world = 1, merge = 0;
  while (argc > 2 && argv[1][0] == '-' && argv[1][1] == '-') { /* --rank r --world n | --merge n */
    if (!strcmp(argv[1], "--rank")) rank = atoi(argv[2]);
    else if (!strcmp(argv[1], "--world")) world = atoi(argv[2]);
    else if (!strcmp(argv[1], "--merge")) merge = atoi(argv[2]);
    else break;
    argv[2] = argv[0];
    argv += 2;
    argc -= 2;
  }
  if (argc < 5 || world < 1 || (rank >= 0 && rank >= world)) {
    fprintf(stderr, "usage: %s [--rank r --world n | --merge n] in.bam ref.fa out.bcf report.json [sample]\n", argv[0]);
    return 2;
  }
  const char *sample = argc > 5 ? argv[5] : "SAMPLE";
  if (merge > 0) return merge_main(merge, argv, sample);
  const int sharded = rank >= 0;
  const int host_prep = getenv("BAM2BCF_HOST_PREP") != NULL;
  const int host_bcf = host_prep || getenv("BAM2BCF_HOST_BCF") != NULL;
  const int host_reader = host_bcf || getenv("BAM2BCF_HOST_READER") != NULL;
  if (sharded && host_reader) {
    fprintf(stderr, "bam2bcf: a sharded run reads through the device reader (its contig selection)\n");
    return 2;
  }
  const double t_start = now();
  bsc_params prm = {0.01, 0.05, 2.0, 20, 0};
  bsc_context *ctx;
  CHECK(bsc_create(&prm, &ctx));
  const double t_ctx = now() - t_start;
  bsc_bam *bam = NULL;
  bsc_bamdev *dev = NULL;
  if (host_reader) CHECK(bsc_bam_open_threads(argv[1], getenv("BAM2BCF_THREADS") ? atoi(getenv("BAM2BCF_THREADS")) : 4, &bam)); /* BGZF inflate ahead of the parser */
  else if (!sharded) CHECK(bsc_bamdev_open(ctx, argv[1], getenv("BAM2BCF_THREADS") ? atoi(getenv("BAM2BCF_THREADS")) : 0, &dev));
  else { /* this rank's contigs: the stretches of the file that hold them */
    int nr_, *owner;
    const char **nm_;
    uint32_t *ln_;
    bsc_bamstream *hs_;
    header_of(argv[1], &nr_, &nm_, &ln_, &hs_);
    owner = assign_contigs(ln_, nr_, world);
    int32_t *mine = calloc((size_t)nr_ + 2, sizeof *mine);
    uint32_t n_mine = 0;
    for (int i = 0; i < nr_; i++)
      if (owner[i] == rank) mine[n_mine++] = i;
    if (nr_ && owner[nr_ - 1] == rank) mine[n_mine++] = -1; /* the unplaced reads at the file's end go with the last contig's rank */
    CHECK(bsc_bamdev_open_contigs(ctx, argv[1], getenv("BAM2BCF_THREADS") ? atoi(getenv("BAM2BCF_THREADS")) : 0, mine, n_mine, &dev));
    free(mine);
    free(owner);
    free(nm_);
    free(ln_);
    bsc_bamstream_close(hs_);
  }
#define N_REFS() (host_reader ? bsc_bam_n_refs(bam) : bsc_bamdev_n_refs(dev))
#define REF_NAME(i) (host_reader ? bsc_bam_ref_name(bam, (i)) : bsc_bamdev_ref_name(dev, (i)))
#define REF_LEN(i) (host_reader ? bsc_bam_ref_len(bam, (i)) : bsc_bamdev_ref_len(dev, (i)))
  FILE *out = sharded ? NULL : fopen(argv[3], "wb");
  if (!out && !sharded) {
    perror(argv[3]);
    return 1;
  }
  const int n_ref = N_REFS();
  if (!sharded) {
    const char **names = calloc((size_t)n_ref + 1, sizeof *names);
    uint32_t *lens = calloc((size_t)n_ref + 1, sizeof *lens);
    for (int i = 0; i < n_ref; i++) {
      names[i] = REF_NAME(i);
      lens[i] = REF_LEN(i);
    }
    write_header(out, n_ref, names, lens, sample);
    free(names);
    free(lens);
  }
  bsc_bcf_ids ids;
  bsc_bcf_default_ids(&ids);
  const bsc_reader_params rpar = {20, 1000, 0, 0, 0, 0, 0, 0}; /* defaults of the reference; no region */
  const bsc_prep_params ppar = {{0, 0}, {0, 0}, 20};

  bsc_contig_totals *ctot = calloc((size_t)n_ref + 1, sizeof *ctot);
  static uint64_t prof_counts[PROF_CAP][4];
  memset(prof_counts, 0, sizeof prof_counts);
  bsc_read_profile prof = {NULL, 0, 0, &prof_counts[0][0], 4096, 0};
  uint64_t base_filter[5] = {0, 0, 0, 0, 0}, passed_reads = 0, passed_bases = 0, before[14], after[14];
  CHECK(bsc_reset_site_stats(ctx));
  CHECK(bsc_get_site_totals(ctx, before));

  uint8_t *codes = NULL, *ref = NULL, *pseq = NULL, *bcf = NULL, *gc = NULL;
  bsc_template *tpl = NULL;
  bsc_vcf_rec *recs = NULL;
  uint64_t codes_len = 0;
  size_t cap_ref = 0, cap_pseq = 0, cap_tpl = 0, cap_recs = 0, cap_bcf = 0;
  int cur_tid = -1;
  uint64_t n_blocks = 0, n_records = 0;
  bsc_read_block blk;
  bsc_dev_read_block dblk;
  static out_writer W;
  static ref_job RJ;
  int out_fd = -1;
  uint64_t file_at = 0;
  if (n_ref > 0) ref_start(&RJ, argv[2], REF_NAME(0), REF_LEN(0), 0); /* the first contig's reference: while the helpers inflate */
  if (!host_reader) { /* the output pieces too: page-locking them takes its time */
    if (out) {
      fflush(out);
      file_at = (uint64_t)ftello(out);
      out_fd = fileno(out);
    } /* else a rank of a sharded run: a file per contig, opened when the contig begins */
    W.ctx = ctx;
    for (int k = 0; k < 2; k++) W.buf[k] = pinned(PIECE);
    pthread_mutex_init(&W.mu, NULL);
    pthread_cond_init(&W.cv, NULL);
    pthread_create(&W.th, NULL, writer_main, &W);
    W.started = 1;
  }
  int r;
  double t_read = 0, t_ref = 0, t_prep = 0, t_gpu = 0, t_enc = 0, t0 = now(), t1;
  for (;;) {
    if (host_reader) r = bsc_bam_next_block(bam, &rpar, &blk);
    else {
      r = bsc_bamdev_next_block(dev, &rpar, &dblk);
      blk.tid = dblk.tid; /* the loop below reads the block's contig, end and first position from blk / x */
      blk.y = dblk.y;
      blk.nr = dblk.nr;
    }
    t_read += (t1 = now()) - t0;
    t0 = t1;
    if (r != 1) break;
    if (blk.tid != cur_tid) { /* contig change: its sequence, and the finished contig's share of the totals */
      if (cur_tid >= 0) {
        CHECK(bsc_get_site_totals(ctx, after));
        uint64_t *d = ctot[cur_tid].snps; /* seven [all, passed] pairs, contiguous */
        for (int i = 0; i < 14; i++) d[i] += after[i] - before[i];
        memcpy(before, after, sizeof before);
      }
      cur_tid = blk.tid;
      ctot[cur_tid].name = REF_NAME(cur_tid);
      if (sharded) { /* this contig's shard; the one before is closed by the output thread behind its last piece */
        if (out_fd >= 0) {
          const out_job cj = {NULL, 0, 0, out_fd, 1};
          writer_push(&W, cj);
        }
        char *sp = malloc(strlen(argv[3]) + 32);
        sprintf(sp, "%s.shard%05d", argv[3], cur_tid);
        FILE *sf = fopen(sp, "wb");
        if (!sf) {
          perror(sp);
          return 1;
        }
        out_fd = dup(fileno(sf));
        fclose(sf);
        free(sp);
        file_at = 0;
      }
      /* its sequence and GC bins (for the report's GC-by-coverage table: load_sequence computes them when a report is asked for): loaded in
       * the background since the previous contig began — or since the program did */
      if (!RJ.running || RJ.tid != cur_tid) {
        if (RJ.running) {
          pthread_join(RJ.th, NULL);
          free(RJ.codes);
          free(RJ.gc);
        }
        ref_start(&RJ, argv[2], REF_NAME(cur_tid), REF_LEN(cur_tid), cur_tid);
      }
      if (RJ.running) pthread_join(RJ.th, NULL);
      RJ.running = 0;
      if (RJ.rc < 0) {
        fprintf(stderr, "reference of %s: %s\n", REF_NAME(cur_tid), RJ.err);
        return 1;
      }
      free(codes);
      free(gc);
      codes = RJ.codes;
      gc = RJ.gc;
      codes_len = RJ.codes_len;
      CHECK(bsc_set_gc_bins_host(ctx, gc, (uint32_t)RJ.n_bins, RJ.gc_start));
      if (cur_tid + 1 < n_ref) ref_start(&RJ, argv[2], REF_NAME(cur_tid + 1), REF_LEN(cur_tid + 1), cur_tid + 1); /* the next one, meanwhile */
    }
    const uint32_t x = host_reader ? bsc_block_start(&blk.tpl[0]) : dblk.x, y = blk.y, n = y - x + 1;
    if (n + 2 > cap_ref) ref = xrealloc(ref, cap_ref = (size_t)(n + 2) * 2);
    CHECK(bsc_block_reference(codes, codes_len, x, n + 2, ref));
    t_ref += (t1 = now()) - t0;
    t0 = t1;
    /* the process thread's per-template work and call_genotypes_ML in one call: the raw templates go up as the reader left them,
     * the device prepares them (trims, clips, mate overlap, indels; the read profile) and calls the block.  BAM2BCF_HOST_PREP in the
     * environment keeps round 4's split (bsc_prepare_templates_profile on this thread, then bsc_block_records) for comparison. */
    bsc_prep_stats st;
    const bsc_vcf_params vp = {0, 1, (uint32_t)codes_len};
    uint64_t n_out = 0, n_bytes = 0;
    if (!host_reader) { /* the block is in HBM already; its stream stays there too and comes over in pieces, the output thread writing behind */
      uint64_t dev_cap = (uint64_t)n * 96 + 4096;
      int rc = bsc_block_bcf_rawdev_keep(ctx, dblk.d_tpl, dblk.nr, dblk.d_seq, dblk.seq_bytes, dblk.d_misms, dblk.n_misms, dblk.ins_pad, &ppar, x, y, ref, NULL,
                                         &vp, 1, dblk.tid, &ids, NULL, dev_cap, &n_bytes, &n_out, &st, &prof);
      if (rc == BSC_ERR_ARG && n_bytes > dev_cap) { /* a block of long records: the encoder alone once more, with the room it asks for */
        dev_cap = n_bytes + 4096;
        rc = bsc_block_bcf_again(ctx, NULL, dev_cap, &n_bytes, &n_out);
      }
      CHECK(rc);
      t_gpu += (t1 = now()) - t0;
      t0 = t1;
      if (n_bytes) { /* the stream changes hands: the output thread reads it out and writes it while this one goes on to the next block */
        void *d_stream = NULL;
        uint64_t n_det = 0;
        CHECK(bsc_bcf_stream_detach(ctx, &d_stream, &n_det));
        const out_job j = {d_stream, n_det, file_at, out_fd, 0};
        writer_push(&W, j);
      }
      file_at += n_bytes;
      n_bytes = 0; /* written by the output thread */
    } else if (!host_bcf) { /* the whole block on the device, the encoding included.  Room for 96 bytes per position: a WGBS block writes a
                      * record of ~113 bytes for every second position; a block that needs more says so and is encoded again */
      if ((size_t)n * 96 + 4096 > cap_bcf) {
        bsc_free_host(bcf);
        bcf = pinned(cap_bcf = (size_t)n * 96 + 4096);
      }
      int rc = bsc_block_bcf_raw(ctx, blk.tpl, blk.nr, blk.seq, blk.seq_bytes, blk.misms, blk.n_misms, &ppar, x, y, ref, NULL, &vp, 1, blk.tid, &ids, NULL, bcf,
                                 cap_bcf, &n_bytes, &n_out, &st, &prof);
      if (rc == BSC_ERR_ARG && n_bytes > cap_bcf) { /* a block of long records: the encoder alone once more, with the room it asks for */
        bsc_free_host(bcf);
        bcf = pinned(cap_bcf = (size_t)n_bytes + 4096);
        rc = bsc_block_bcf_again(ctx, bcf, cap_bcf, &n_bytes, &n_out);
      }
      CHECK(rc);
    } else if (!host_prep) {
      if (n > cap_recs) recs = xrealloc(recs, (cap_recs = (size_t)n * 2) * sizeof *recs);
      CHECK(bsc_block_records_raw(ctx, blk.tpl, blk.nr, blk.seq, blk.seq_bytes, blk.misms, blk.n_misms, &ppar, x, y, ref, NULL, &vp, 1, recs,
                                  cap_recs, &n_out, &st, &prof));
    } else {
      if (n > cap_recs) recs = xrealloc(recs, (cap_recs = (size_t)n * 2) * sizeof *recs);
      uint64_t pad = 0;
      for (uint64_t i = 0; i < blk.n_misms; i++)
        if (blk.misms[i].type == BSC_MISMS_INS) pad += blk.misms[i].size;
      if (blk.seq_bytes + pad + 16 > cap_pseq) pseq = xrealloc(pseq, cap_pseq = (size_t)(blk.seq_bytes + pad + 16) * 2);
      if (blk.nr > cap_tpl) tpl = xrealloc(tpl, (cap_tpl = (size_t)blk.nr * 2) * sizeof *tpl);
      uint64_t used = 0;
      prof.ref = ref;
      prof.x = x;
      prof.n_ref = n + 2;
      CHECK(bsc_prepare_templates_profile(blk.tpl, blk.nr, blk.seq, blk.seq_bytes, blk.misms, blk.n_misms, &ppar, tpl, pseq, cap_pseq, &used, &st,
                                          &prof));
      t_prep += (t1 = now()) - t0;
      t0 = t1;
      CHECK(bsc_block_records(ctx, tpl, blk.nr, pseq, used, x, y, ref, NULL, &vp, 1, recs, cap_recs, &n_out));
    }
    base_filter[0] += st.base_none;
    base_filter[1] += st.base_trim;
    base_filter[2] += st.base_clip;
    base_filter[3] += st.base_overlap;
    base_filter[4] += st.base_lowqual;
    passed_reads += st.reads;
    passed_bases += st.read_bases;
    t_gpu += (t1 = now()) - t0;
    t0 = t1;
    if (host_bcf) {
      if (n_out * 256 + 64 > cap_bcf) {
        bsc_free_host(bcf);
        bcf = pinned(cap_bcf = (size_t)(n_out * 256 + 64) * 2);
      }
      uint64_t done = 0;
      const long nb = bsc_bcf_block(recs, n_out, blk.tid, &ids, NULL, bcf, cap_bcf, &done);
      CHECK(nb);
      if (done != n_out) {
        fprintf(stderr, "BCF buffer too small\n");
        return 1;
      }
      n_bytes = (uint64_t)nb;
    }
    if (n_bytes) fwrite(bcf, 1, (size_t)n_bytes, out);
    n_blocks++;
    n_records += n_out;
    t_enc += (t1 = now()) - t0;
    t0 = t1;
  }
  CHECK(r);
  if (W.started) { /* the output thread writes what it still holds, then goes */
    pthread_mutex_lock(&W.mu);
    W.quit = 1;
    pthread_cond_broadcast(&W.cv);
    pthread_mutex_unlock(&W.mu);
    pthread_join(W.th, NULL);
    if (W.failed) {
      fprintf(stderr, "writing %s failed\n", argv[3]);
      return 1;
    }
    t_enc += (t1 = now()) - t0;
    t0 = t1;
  }
  const double t_loop_end = now();
  if (cur_tid >= 0) {
    CHECK(bsc_get_site_totals(ctx, after));
    uint64_t *d = ctot[cur_tid].snps;
    for (int i = 0; i < 14; i++) d[i] += after[i] - before[i];
  }
  if (out) fclose(out);
  else if (out_fd >= 0) close(out_fd);

  /* the report */
  static bsc_site_stats total;
  CHECK(bsc_get_site_stats(ctx, &total));
  bsc_report rep;
  memset(&rep, 0, sizeof rep);
  rep.under_conv = prm.under_conv;
  rep.over_conv = prm.over_conv;
  rep.mapq_thresh = 20;
  rep.min_qual = 20;
  rep.day = 1, rep.month = 1, rep.year = 2000; /* a fixed date: reproducible output */
  if (host_reader) bsc_bam_filter_counts(bam, rep.filter_cts, rep.filter_bases);
  else CHECK(bsc_bamdev_filter_counts(dev, rep.filter_cts, rep.filter_bases));
  const unsigned long long malformed = host_reader ? bsc_bam_malformed(bam) : bsc_bamdev_malformed(dev);
  if (malformed)
    fprintf(stderr, "bam2bcf: warning: %llu BAM records dropped, their CIGAR does not cover the sequence (damaged input?)\n", malformed);
  rep.filter_cts[0] += passed_reads;
  rep.filter_bases[0] += passed_bases;
  memcpy(rep.base_filter, base_filter, sizeof base_filter);
  rep.total = &total;
  CHECK(bsc_set_gc_bins_host(ctx, NULL, 0, 0));
  uint64_t *gc_table = xrealloc(NULL, (size_t)BSC_COV_CAP * 101 * sizeof(uint64_t));
  CHECK(bsc_get_gc_stats(ctx, gc_table));
  rep.gc = gc_table;
  rep.read_profile = &prof_counts[0][0];
  rep.n_read_profile = prof.used;
  /* contigs in header order, as the reference lists them */
  bsc_contig_totals *listed = calloc((size_t)n_ref + 1, sizeof *listed);
  uint32_t nl = 0;
  for (int i = 0; i < n_ref; i++)
    if (ctot[i].name) listed[nl++] = ctot[i];
  rep.contigs = listed;
  rep.n_contigs = nl;
  if (sharded) { /* this rank's share of every sum, for bam2bcf --merge */
    rank_sums *rs_ = calloc(1, sizeof *rs_);
    rs_->magic = SUMS_MAGIC;
    rs_->n_ref = (uint64_t)n_ref;
    rs_->n_blocks = n_blocks;
    rs_->n_records = n_records;
    rs_->malformed = malformed;
    memcpy(rs_->filter_cts, rep.filter_cts, sizeof rs_->filter_cts);
    memcpy(rs_->filter_bases, rep.filter_bases, sizeof rs_->filter_bases);
    memcpy(rs_->base_filter, base_filter, sizeof rs_->base_filter);
    rs_->prof_used = prof.used;
    memcpy(rs_->prof, prof_counts, sizeof rs_->prof);
    rs_->total = total;
    uint64_t *ct = calloc((size_t)n_ref * 15 + 1, 8);
    for (int i = 0; i < n_ref; i++)
      if (ctot[i].name) {
        ct[(size_t)i * 15] = 1;
        memcpy(ct + (size_t)i * 15 + 1, ctot[i].snps, 14 * 8);
      }
    char *sp = malloc(strlen(argv[3]) + 32);
    sprintf(sp, "%s.rank%05d.sums", argv[3], rank);
    FILE *fs = fopen(sp, "wb");
    if (!fs || fwrite(rs_, sizeof *rs_, 1, fs) != 1 || fwrite(gc_table, 8, (size_t)BSC_COV_CAP * 101, fs) != (size_t)BSC_COV_CAP * 101 ||
        fwrite(ct, 8, (size_t)n_ref * 15, fs) != (size_t)n_ref * 15 || fclose(fs)) {
      perror(sp);
      return 1;
    }
    printf("rank %d of %d: %llu blocks, %llu records written\n", rank, world, (unsigned long long)n_blocks, (unsigned long long)n_records);
  } else {
    const long need = bsc_report_json(&rep, NULL, 0);
    CHECK(need);
    char *text = xrealloc(NULL, (size_t)need + 1);
    bsc_report_json(&rep, text, (size_t)need + 1);
    FILE *fr = fopen(argv[4], "w");
    if (!fr) {
      perror(argv[4]);
      return 1;
    }
    fwrite(text, 1, (size_t)need, fr);
    fclose(fr);
    printf("%llu blocks, %llu records written\n", (unsigned long long)n_blocks, (unsigned long long)n_records);
  }
  if (getenv("BAM2BCF_TIMING")) {
    uint64_t rc4[4] = {0, 0, 0, 0};
    double rs[2] = {0, 0};
    if (dev) bsc_bamdev_run_stats(dev, rc4, rs);
    fprintf(stderr,
            "{\"reader\": \"%s\", \"context_s\": %.3f, \"reader_s\": %.3f, \"reference_s\": %.3f, \"host_prep_s\": %.3f, \"block_call\": \"%s\", "
            "\"block_call_s\": %.3f, \"encode_write_s\": %.3f, \"report_s\": %.3f, \"wall_s\": %.3f, \"wall_without_context_s\": %.3f, "
            "\"device_reader\": {\"passes\": %llu, \"replay_passes\": %llu, \"records\": %llu, \"inflated_bytes\": %llu, \"waiting_for_inflate_s\": %.3f, "
            "\"device_passes_s\": %.3f}, \"output_thread\": {\"waiting_for_a_block_s\": %.3f, \"waiting_for_copies_s\": %.3f, \"pwrite_s\": %.3f}}\n",
            host_reader ? "host (csrc/bamio.c)" : "device (csrc/bamstream.c + csrc/bamdev.hip)", t_ctx, t_read, t_ref, t_prep,
            host_prep ? "bsc_block_records" : (host_bcf ? "bsc_block_records_raw" : (host_reader ? "bsc_block_bcf_raw" : "bsc_block_bcf_rawdev")), t_gpu, t_enc,
            now() - t_loop_end, now() - t_start, now() - t_start - t_ctx, (unsigned long long)rc4[0], (unsigned long long)rc4[1], (unsigned long long)rc4[2],
            (unsigned long long)rc4[3], rs[0], rs[1], W.t_idle, W.t_wait, W.t_write);
  }
  if (RJ.running) { /* a contig prefetched and never reached */
    pthread_join(RJ.th, NULL);
    free(RJ.codes);
    free(RJ.gc);
  }
  if (bam) bsc_bam_close(bam);
  if (dev) bsc_bamdev_close(dev);
  bsc_destroy(ctx);
  return 0;
}
