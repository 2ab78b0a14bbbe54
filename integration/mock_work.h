/*
 * mock_work.h — a mock of the fields of the reference's work_t (include/bs_call.h:230-282) that the replacement
 * call_genotypes_ML touches, with the three threads around it written as the reference writes them: the print thread
 * (src/process.c:74-110), the meth profiling thread (src/process.c:20-45 over src/meth_profile.c:48-77, which reads
 * work->ref1) and the process thread's side of both (src/process_template.c:29-30,116-124: it overwrites ref1 for the next
 * block and queues one profiling job per template).  Used by integration/demo_block.c (with the GPU library) and
 * integration/overlap_tsan.c (CPU only, stub library, ThreadSanitizer).
 */
#ifndef MOCK_WORK_H
#define MOCK_WORK_H

#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <bscall_amd.h>

/* `gt_vcf`, include/bs_call.h:162-166: what the calc side publishes per position (208 bytes) */
typedef struct {
  bsc_gt_meth gtm;
  bool ready;
  bool skip;
} gt_vcf;
_Static_assert(sizeof(gt_vcf) == 208, "gt_vcf is 208 bytes: the out_stride of the gt_vcf[] form");

#define N_MPROF_BUFFERS 256 /* include/bs_call.h:228 */
typedef struct {
  uint32_t off, len; /* the stretch of the block's reference codes this job looks at (a template's span) */
  uint64_t expect;   /* its checksum as the process thread wrote the codes */
} mprof_job_t;

typedef struct mock_ctg mock_ctg; /* opaque, like ctg_t for the protocol */

typedef struct {
  gt_vcf *vcf;
  int vcf_size, vcf_n;
  uint32_t vcf_x;
  mock_ctg *vcf_ctg;
  bool print_end, mprof_end;
  pthread_mutex_t print_mutex, vcf_mutex, mprof_mutex;
  pthread_cond_t print_cond1, print_cond2, vcf_cond, mprof_cond1, mprof_cond2;
  mprof_job_t mprof_thread[N_MPROF_BUFFERS];
  int mprof_read_idx, mprof_write_idx;
  char *ref, *ref1; /* gt_strings in the reference: the codes of the block being printed / being handed over */
  size_t ref_cap, ref1_cap;
  /* what the mock's threads keep */
  uint64_t hash, records, covered; /* print thread: every byte it would read */
  uint64_t ref_hash;               /* print thread: the reference codes it saw beside each block */
  uint64_t mprof_jobs, mprof_bad;  /* profiling thread: jobs done, jobs that found ref1 changed under them */
  uint64_t bcf_hash, bcf_bytes;    /* the writer of the bytes form (amd_bcf_protocol.h): every byte it was given, in order */
  long print_ns;                   /* BSC_DEMO_PRINT_NS: what the mock printer spends per position on top of hashing it (the
                                      reference's prints a record in about a microsecond) — to time the hand-over under a slow consumer */
} work_t;

static uint64_t mock_fnv(uint64_t h, const void *p, size_t n) {
  const unsigned char *b = p;
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

static void mock_consume(work_t *w, const gt_vcf *v) { /* stands in for print_vcf_entry: every byte the printer would read */
  if (w->print_ns < 0) { /* BSC_DEMO_PRINT_NS < 0: a consumer that only counts (timings of the hand-over itself; no hash to compare) */
    w->records++;
    w->covered += !v->skip;
    return;
  }
  uint64_t q[sizeof v->gtm / 8], h = w->hash;
  memcpy(q, &v->gtm, sizeof q);
  for (size_t i = 0; i < sizeof q / sizeof q[0]; i++) h = (h ^ q[i]) * 1099511628211ull;
  h = (h ^ (unsigned)v->skip) * 1099511628211ull;
  w->hash = h;
  w->records++;
  w->covered += !v->skip;
}

/* print_thread, src/process.c:74-110: wait for a block (vcf_n > 0), take its positions in index order as their `ready`
 * flags appear (reading work->ref beside them), then declare the block drained (vcf_n = 0, print_cond2) */
static void *mock_print_thread(void *arg) {
  work_t *w = arg;
  for (;;) {
    pthread_mutex_lock(&w->print_mutex);
    while (!w->vcf_n && !w->print_end) pthread_cond_wait(&w->print_cond1, &w->print_mutex);
    const int n = w->vcf_n;
    pthread_mutex_unlock(&w->print_mutex);
    if (!n) break;
    w->ref_hash = mock_fnv(w->ref_hash, w->ref, (size_t)n + 2);
    for (int i = 0; i < n; i++) {
      gt_vcf *v = w->vcf + i;
      if (!__atomic_load_n(&v->ready, __ATOMIC_ACQUIRE)) {
        pthread_mutex_lock(&w->vcf_mutex);
        while (!__atomic_load_n(&v->ready, __ATOMIC_ACQUIRE)) pthread_cond_wait(&w->vcf_cond, &w->vcf_mutex);
        pthread_mutex_unlock(&w->vcf_mutex);
      }
      mock_consume(w, v);
    }
    if (w->print_ns > 0) {
      const long long ns = (long long)w->print_ns * n;
      struct timespec ts = {(time_t)(ns / 1000000000ll), (long)(ns % 1000000000ll)};
      nanosleep(&ts, NULL);
    }
    pthread_mutex_lock(&w->print_mutex);
    w->vcf_n = 0;
    pthread_cond_signal(&w->print_cond2);
    pthread_mutex_unlock(&w->print_mutex);
  }
  return NULL;
}

/* mprof_thread, src/process.c:20-45: take jobs off the ring; each reads work->ref1 as meth_profile does (src/meth_profile.c:51) */
static void *mock_mprof_thread(void *arg) {
  work_t *w = arg;
  pthread_mutex_lock(&w->mprof_mutex);
  for (;;) {
    while (w->mprof_read_idx == w->mprof_write_idx && !w->mprof_end) pthread_cond_wait(&w->mprof_cond1, &w->mprof_mutex);
    const bool end = w->mprof_read_idx == w->mprof_write_idx;
    const int ix = w->mprof_read_idx;
    pthread_mutex_unlock(&w->mprof_mutex);
    if (end) break;
    const mprof_job_t *j = &w->mprof_thread[ix];
    const uint64_t h = mock_fnv(1469598103934665603ull, w->ref1 + j->off, j->len);
    w->mprof_jobs++;
    w->mprof_bad += h != j->expect;
    pthread_mutex_lock(&w->mprof_mutex);
    w->mprof_read_idx = (ix + 1) % N_MPROF_BUFFERS;
    pthread_cond_broadcast(&w->mprof_cond2);
  }
  return NULL;
}

static void mock_work_init(work_t *w) {
  memset(w, 0, sizeof *w);
  w->hash = w->ref_hash = 1469598103934665603ull;
  {
    const char *e = getenv("BSC_DEMO_PRINT_NS");
    w->print_ns = e && *e ? atol(e) : 0;
  }
  pthread_mutex_init(&w->print_mutex, NULL);
  pthread_mutex_init(&w->vcf_mutex, NULL);
  pthread_mutex_init(&w->mprof_mutex, NULL);
  pthread_cond_init(&w->print_cond1, NULL);
  pthread_cond_init(&w->print_cond2, NULL);
  pthread_cond_init(&w->vcf_cond, NULL);
  pthread_cond_init(&w->mprof_cond1, NULL);
  pthread_cond_init(&w->mprof_cond2, NULL);
}

/* The process thread's part before a call (src/process_template.c:29-30,116-124): the block's reference codes go into
 * work->ref1 — overwriting the previous block's, which is only safe because the previous call waited for the profiling
 * thread — and `jobs` profiling jobs are queued. */
static void mock_prepare_block(work_t *w, const uint8_t *ref, uint32_t sz, int jobs) {
  if ((size_t)sz + 3 > w->ref1_cap) { /* gt_string_resize(work->ref1, sz + 3) */
    w->ref1 = realloc(w->ref1, (size_t)sz + 3);
    w->ref1_cap = (size_t)sz + 3;
  }
  memcpy(w->ref1, ref, (size_t)sz + 2);
  w->ref1[sz + 2] = 0;
  for (int k = 0; k < jobs; k++) {
    mprof_job_t *mp = &w->mprof_thread[w->mprof_write_idx];
    mp->len = sz + 2 < 400u ? sz + 2 : 400u;
    mp->off = (uint32_t)(((uint64_t)k * 7919u) % (sz + 3 - mp->len));
    mp->expect = mock_fnv(1469598103934665603ull, ref + mp->off, mp->len);
    const int ix = (w->mprof_write_idx + 1) % N_MPROF_BUFFERS;
    pthread_mutex_lock(&w->mprof_mutex);
    while (ix == w->mprof_read_idx) pthread_cond_wait(&w->mprof_cond2, &w->mprof_mutex);
    w->mprof_write_idx = ix;
    pthread_cond_signal(&w->mprof_cond1);
    pthread_mutex_unlock(&w->mprof_mutex);
  }
}

/* the glue's view of the mock (integration/amd_overlap_protocol.h) */
#define AMD_WORK_T work_t
#define AMD_GT_VCF_T gt_vcf
#define AMD_CTG_T mock_ctg
#define AMD_SET_REF(work, src, sz)                                       \
  do {                                                                   \
    if ((size_t)(sz) + 3 > (work)->ref_cap) {                            \
      (work)->ref = realloc((work)->ref, (size_t)(sz) + 3);              \
      (work)->ref_cap = (size_t)(sz) + 3;                                \
    }                                                                    \
    memcpy((work)->ref, (src), (size_t)(sz) + 3);                        \
  } while (0)
#define AMD_REF1(work) ((const char *)(work)->ref1)

#endif /* MOCK_WORK_H */
