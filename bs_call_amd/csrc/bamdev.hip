/*
 * bamdev.hip — BAM records -> blocks of raw templates ON THE DEVICE (round 6): row f-4's reader, MI355X-first.  The host inflates BGZF
 * blocks into page-locked slabs and finds the record boundaries (csrc/bamstream.c); everything the reference's reader thread does with a
 * record happens here, on the bytes as they lie in HBM:
 *
 *   bsc_bam_parse_kernel      a lane per record: get_next_align_details (src/input_sam.c:222-312) -> a 64-byte descriptor — flag / MAPQ /
 *                             insert / orientation filters with their fifteen counters, forward / reverse position, the CIGAR's span
 *                             (:90-136), the strand tag (:144-220), the name's hash
 *   used-record compaction, then read_input (src/get_template_vector.c:49-389) as independent pieces (csrc/bamdev_core.h, bd_f_*):
 *     blocks                  ONE inclusive max-scan of (contig run | rightmost covered position) + a lane per record (:141-207)
 *     names                   rocPRIM radix sort of (contig run ^ name hash): a name's records side by side (:223-276)
 *     duplicates              a lane per start position walks its records through the reference's statements (:281-326,345-372)
 *     mates                   a lane per backwards-facing mate joins its partner's template
 *   bsc_bam_replay_kernel     the same statements by ONE lane over all records, for input the pieces are not exact for (re-used names,
 *                             unsorted records, mates that disagree ...): slow, exact, and the one that names an error
 *   assembly                  prefix sums place reads, lists and templates; a wave per record decodes nibbles + qualities into
 *                             base | min(q, 43) << 2 bytes (src/input_sam.c:61-88); bsc_raw_template[] per block, offsets block-relative
 *
 * Output = exactly what bsc_prepare_templates_device / bsc_block_bcf_rawdev take, in HBM: nothing crosses PCIe twice.  The statements are
 * csrc/bamdev_core.h's, which tests/emul runs on the CPU against csrc/bamio.c and the test suite's independent Python restatement; tests/test_gpu_bamdev.py checks these
 * kernels' bytes against both.
 */
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/bscall_amd.h"
#include "bamdev_core.h"

extern "C" int bsc_set_error(int code, const char *fmt, ...);
extern "C" int bsc_ctx_device(const bsc_context *ctx);
extern "C" void *bsc_ctx_stream(const bsc_context *ctx);

/* ---- kernels ------------------------------------------------------------------------------------------------------------------ */
struct bam_cnt { /* device counters of the reader */
  unsigned long long cts[15], bases[15]; /* filter_cts / filter_bases */
  unsigned long long malformed;
  unsigned long long first_err; /* lowest record index with an error status, ~0 = none */
  unsigned long long err[2];    /* the replay's BD_E_* and the used record it names */
  uint32_t irregular, u_done, n_done, n_groups;
  uint32_t n_blocks, n_tpl, blk_err, pad_;
  unsigned long long seq_total, ms_total;
};

__global__ void bsc_bam_parse_kernel(const uint8_t *arena, uint64_t arena_base, uint64_t arena_end, uint64_t base_off, const uint32_t *rel, uint32_t n,
                                     bd_params par, bd_desc *out, uint32_t out0, bam_cnt *cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t off = base_off + rel[i];
  bd_desc d;
  bd_parse(arena + (off - arena_base), arena_end - off, off, par, d);
  out[out0 + i] = d;
  if (d.status == BD_ST_FILTERED) {
    atomicAdd(&cnt->cts[d.flt], 1ull);
    atomicAdd(&cnt->bases[d.flt], (unsigned long long)d.l_seq);
  } else if (d.status == BD_ST_MALFORMED_CIGAR)
    atomicAdd(&cnt->malformed, 1ull);
  else if (d.status >= BD_ST_ERR_SIZE)
    atomicMin(&cnt->first_err, (unsigned long long)(out0 + i));
}

__global__ void bsc_bam_flag_used_kernel(const bd_desc *D, uint32_t n, uint32_t *flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) flag[i] = (i < n && D[i].status == BD_ST_USE) ? 1u : 0u;
}
__global__ void bsc_bam_scatter_used_kernel(const uint32_t *flag, const uint32_t *pos, uint32_t n, uint32_t *U) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) U[pos[i]] = i;
}

__global__ void bsc_bam_run_flag_kernel(bd_ws ws, uint32_t *rf) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < ws.n_used) rf[u] = bd_f_run_start(ws, u) ? 1u : 0u;
}
__global__ void bsc_bam_key_kernel(bd_ws ws, const uint32_t *run, unsigned long long *key, unsigned long long *nkey, uint32_t *nval) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= ws.n_used) return;
  const bd_desc &d = BD_D(ws, u);
  key[u] = bd_f_key(run[u], d);
  nkey[u] = (d.aflag & BD_F_PAIRED) ? ((d.hash ^ ((uint64_t)run[u] * 0x9e3779b97f4a7c15ull)) & ~1ull) : ~0ull; /* all ones: not in the table */
  nval[u] = u;
}
__global__ void bsc_bam_open_kernel(bd_ws ws, bd_params par, const unsigned long long *scan, uint32_t *flags, uint32_t *f_blk, uint32_t *f_grp, bam_cnt *cnt) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= ws.n_used) return;
  uint32_t irr = 0;
  const uint32_t f = bd_f_open(ws, par, u, u ? (uint32_t)scan[u - 1] : 0u, &irr);
  flags[u] = f;
  f_blk[u] = f & 1u;
  f_grp[u] = (f >> 1) & 1u;
  ws.blk_open[u] = (uint8_t)(f & 1u);
  ws.max_at[u] = (uint32_t)scan[u];
  if (irr) cnt->irregular = 1u;
  if (f & 1u) atomicMax(&cnt->u_done, u); /* the last block's first record */
}
__global__ void bsc_bam_group_first_kernel(const uint32_t *flags, const uint32_t *grp_of, uint32_t n, uint32_t *g_first) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n && (flags[u] & 2u)) g_first[grp_of[u] - 1u] = u;
}
__global__ void bsc_bam_chain_kernel(bd_ws ws, const unsigned long long *skey, const uint32_t *sval, const uint32_t *blk_of, uint32_t *partner, bam_cnt *cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ws.n_used) return;
  const unsigned long long k = skey[i];
  if (k == ~0ull || (i && skey[i - 1] == k)) return; /* not in the table, or not a chain's first */
  uint32_t n = 1;
  while (n < 3 && i + n < ws.n_used && skey[i + n] == k) n++;
  uint32_t irr = 0;
  bd_f_chain(ws, sval, i, i + n, blk_of, partner, &irr);
  if (irr) cnt->irregular = 1u;
}
__global__ void bsc_bam_group_kernel(bd_ws ws, bd_params par, const uint32_t *g_first, const uint32_t *grp_of, uint32_t u_done, bam_cnt *cnt) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (u_done == 0) return;
  const uint32_t n_groups = grp_of[u_done - 1u]; /* the groups of the complete blocks */
  if (g >= n_groups) return;
  const uint32_t a = g_first[g], b = g + 1u < n_groups ? g_first[g + 1u] : u_done;
  uint32_t irr = 0;
  bd_f_group(ws, par, a, b, 0u, &irr);
  if (irr) cnt->irregular = 1u;
}
__global__ void bsc_bam_join_kernel(bd_ws ws, bd_params par, const uint32_t *partner, const uint32_t *blk_of, const unsigned long long *scan, uint32_t *win0,
                                    uint32_t *win1, uint32_t u_done, bam_cnt *cnt) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= u_done) return;
  uint32_t irr = 0;
  bd_f_join(ws, par, u, partner, blk_of, (const uint64_t *)scan, win0, win1, &irr);
  if (irr) cnt->irregular = 1u;
}
__global__ void bsc_bam_wins_kernel(bd_ws ws, const uint32_t *win0, const uint32_t *win1, uint32_t u_done) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= u_done) return;
  if (win0[s]) ws.side0[s] = win0[s] - 1u;
  if (win1[s]) ws.side1[s] = win1[s] - 1u;
}
__global__ void bsc_bam_replay_kernel(bd_ws ws, bd_params par, int final, bam_cnt *cnt) {
  if (blockIdx.x || threadIdx.x) return;
  uint32_t n_done = 0;
  ws.cts = cnt->cts;
  ws.bases = cnt->bases;
  ws.err = cnt->err;
  const int e = bd_replay(ws, par, 0u, final, &n_done);
  cnt->err[0] = (unsigned long long)e;
  cnt->n_done = n_done;
}

/* assembly: per used record of the complete blocks its read's length, its list's length, the bases its deletions pad, whether it made a
 * template, whether it opens a block; one element more (zeros) so that the exclusive sums end with the totals */
__global__ void bsc_bam_asm_in_kernel(bd_ws ws, uint32_t n_done, unsigned long long *len, unsigned long long *nms, unsigned long long *del, uint32_t *slot,
                                      uint32_t *bop) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u > n_done) return;
  if (u == n_done) {
    len[u] = nms[u] = del[u] = 0ull;
    slot[u] = bop[u] = 0u;
    return;
  }
  const bd_desc &d = BD_D(ws, u);
  len[u] = d.l_seq;
  nms[u] = d.n_ms;
  del[u] = d.del_len;
  slot[u] = ws.slot_made[u] ? 1u : 0u;
  bop[u] = ws.blk_open[u] ? 1u : 0u;
}
__global__ void bsc_bam_block_first_kernel(const uint32_t *bop, const uint32_t *bidx, const uint32_t *slot, const uint32_t *tidx, uint32_t n_done, uint32_t *bstart,
                                           uint32_t *tslot) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_done) return;
  if (bop[u]) bstart[bidx[u]] = u; /* bidx: exclusive sum = the block's number */
  if (slot[u]) tslot[tidx[u]] = u;
}

struct bam_blk { /* the table a pass hands to the host */
  int32_t tid;
  uint32_t y, x0, first_tpl, n_tpl, first_used, end_used, pad_;
  unsigned long long seq0, seq_bytes, ms0, n_ms, ins_pad;
};
__global__ void bsc_bam_block_table_kernel(bd_ws ws, const uint32_t *bstart, uint32_t n_blocks, uint32_t n_done, const uint32_t *tidx, const uint32_t *tslot,
                                           const unsigned long long *seq_off, const unsigned long long *ms_off, const unsigned long long *del_off, bam_blk *tab,
                                           bam_cnt *cnt) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const uint32_t a = bstart[b], e = b + 1u < n_blocks ? bstart[b + 1u] : n_done;
  bam_blk t;
  t.tid = BD_D(ws, a).tid;
  t.y = ws.max_at[e - 1u];
  t.first_tpl = tidx[a];
  t.n_tpl = tidx[e] - tidx[a];
  t.first_used = a;
  t.end_used = e;
  t.pad_ = 0;
  t.seq0 = seq_off[a];
  t.seq_bytes = seq_off[e] - seq_off[a];
  t.ms0 = ms_off[a];
  t.n_ms = ms_off[e] - ms_off[a];
  t.ins_pad = del_off[e] - del_off[a];
  t.x0 = 0;
  if (t.n_tpl) {
    const bd_desc &o = BD_D(ws, ws.occ[tslot[t.first_tpl]]);
    t.x0 = o.fwd ? o.fwd : o.rev;
    if (t.x0 == 0 || t.x0 > t.y) atomicMin(&cnt->blk_err, b); /* what the process thread asserts (src/process_template.c:24-26) */
  }
  tab[b] = t;
}
__global__ void bsc_bam_template_kernel(bd_ws ws, uint32_t n_done, const uint32_t *slot, const uint32_t *tidx, const uint32_t *bidx_incl_minus, const bam_blk *tab,
                                        const unsigned long long *seq_off, const unsigned long long *ms_off, bd_raw_template *out) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_done || !slot[s]) return;
  /* the block of s: bidx (exclusive) counts the openers before s; s itself opens one more if it is an opener */
  const uint32_t b = bidx_incl_minus[s] + (ws.blk_open[s] ? 1u : 0u) - 1u;
  bd_raw_template t;
  bd_template(ws, s, (const uint64_t *)seq_off, (const uint64_t *)ms_off, tab[b].seq0, tab[b].ms0, &t);
  out[tidx[s]] = t;
}
__global__ void bsc_bam_misms_kernel(bd_ws ws, uint32_t n_done, const unsigned long long *ms_off, bd_misms *out) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_done) return;
  const bd_desc &d = BD_D(ws, u);
  if (d.n_ms) bd_misms_of(ws.arena + (d.off - ws.arena_base), d, out + ms_off[u]);
}

/* get_seq_and_qual (src/input_sam.c:61-88): a wave per record, a lane per four bases — two bytes of nibbles and four qualities in,
 * one dword out.  SWAR: nibble -> base code by a 16-entry table held in one 64-bit constant pair; quality clamp per byte. */
#define BAM_DEC_WAVES 4
__device__ static __forceinline__ uint32_t bam_dec4(uint32_t nib16 /* four nibbles, first base in bits 12-15 as the file holds them byte-wise */,
                                                    uint32_t q4) {
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t c4 = (nib16 >> (k * 4)) & 15u; /* k-th base: see the caller's arrangement */
    uint32_t q = (q4 >> (8 * k)) & 0xffu;
    q = q > 43u ? 43u : q;
    /* 1 -> A (0), 2 -> C (1), 4 -> G (2), 8 -> T (3); anything else: N = byte 0 */
    const uint32_t isb = (c4 == 1u) | (c4 == 2u) | (c4 == 4u) | (c4 == 8u);
    const uint32_t base = (c4 >> 1) - (c4 >> 3); /* 1 -> 0, 2 -> 1, 4 -> 2, 8 -> 3 */
    const uint32_t byte = isb ? (base | q << 2) : 0u;
    out |= byte << (8 * k);
  }
  return out;
}
__global__ __launch_bounds__(64 * BAM_DEC_WAVES) void bsc_bam_decode_kernel(bd_ws ws, uint32_t n_done, const unsigned long long *seq_off, uint8_t *out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * BAM_DEC_WAVES + (threadIdx.x >> 6));
  const uint32_t n_waves = gridDim.x * BAM_DEC_WAVES;
  for (uint32_t u = wave; u < n_done; u += n_waves) {
    const bd_desc &d = BD_D(ws, u);
    const uint32_t l_seq = d.l_seq;
    if (!l_seq) continue;
    const uint8_t *rec = ws.arena + (d.off - ws.arena_base);
    const uint8_t *seq4 = rec + 36 + d.l_name + 4u * d.n_cigar, *qual = seq4 + (l_seq + 1u) / 2u;
    uint8_t *dst = out + seq_off[u];
    for (uint32_t i = lane * 4u; i < l_seq; i += 256u) {
      if (i + 4u <= l_seq) {
        uint16_t nb;
        uint32_t q4;
        __builtin_memcpy(&nb, seq4 + (i >> 1), 2);
        __builtin_memcpy(&q4, qual + i, 4);
        /* bytes b0 b1: bases i, i+1 in b0 (high nibble first), i+2, i+3 in b1 */
        const uint32_t b0 = nb & 0xffu, b1 = nb >> 8;
        const uint32_t nib = (b0 >> 4) | (b0 & 15u) << 4 | (b1 >> 4) << 8 | (b1 & 15u) << 12;
        const uint32_t w = bam_dec4(nib, q4);
        __builtin_memcpy(dst + i, &w, 4);
      } else {
        for (uint32_t j = i; j < l_seq; j++) dst[j] = bd_base_byte(seq4, qual, j);
      }
    }
  }
}

/* ---- the reader ----------------------------------------------------------------------------------------------------------------- */
#define BAM_TRY(call)                                                                                                         \
  do {                                                                                                                        \
    hipError_t e_ = (call);                                                                                                   \
    if (e_ != hipSuccess) return bsc_set_error(BSC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct dev_buf {
  void *p = nullptr;
  size_t cap = 0;
};
static int buf_reserve(dev_buf &b, size_t need, bool keep = false, size_t keep_bytes = 0, hipStream_t s = nullptr) {
  if (need <= b.cap) return BSC_OK;
  size_t nc = need + need / 4 + 4096;
  void *q = nullptr;
  if (hipMalloc(&q, nc) != hipSuccess) {
    (void)hipGetLastError();
    nc = need;
    if (hipMalloc(&q, nc) != hipSuccess) {
      (void)hipGetLastError();
      return bsc_set_error(BSC_ERR_NOMEM, "device BAM reader: out of device memory (%zu bytes)", need);
    }
  }
  if (keep && b.p && keep_bytes) {
    if (hipMemcpyAsync(q, b.p, keep_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      (void)hipFree(q);
      return bsc_set_error(BSC_ERR_HIP, "device BAM reader: copy into a larger buffer failed");
    }
  }
  if (b.p) (void)hipFree(b.p);
  b.p = q;
  b.cap = nc;
  return BSC_OK;
}
static void buf_free(dev_buf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

struct bsc_bamdev {
  bsc_context *ctx = nullptr;
  int device = 0;
  hipStream_t s = nullptr;
  bsc_bamstream *bs = nullptr;
  bd_params par;
  bool par_set = false;
  uint64_t pass_bytes = 256ull << 20;
  /* the inflated stream in HBM: arena[0] is stream offset arena_base; [live_off, arena_end) is still needed */
  dev_buf arena;
  uint64_t arena_base = 0, arena_end = 0, live_off = 0;
  /* descriptors of the records from the block in hand on */
  dev_buf desc, desc2;
  uint32_t n_desc = 0;
  bool has_pend = false; /* a record whose bytes had not all arrived when its pass ended: parsed by the next */
  uint64_t pend_off = 0;
  dev_buf recoff; /* a pass's record starts (u32, slab-relative), slab after slab */
  dev_buf cnt;    /* bam_cnt */
  dev_buf zero;   /* one zero dword */
  dev_buf ws_mem; /* the pass's arrays, carved from one allocation */
  dev_buf tmp;    /* rocPRIM scratch */
  dev_buf d_tpl, d_seq, d_ms, d_tab;
  dev_buf d_keep; /* the contig selection, a byte per contig */
  bool keep_unplaced = false;
  std::vector<bam_blk> blocks;
  size_t next_blk = 0;
  bool stream_end = false, finished = false;
  unsigned long long cts[15] = {0}, bases[15] = {0}, malformed = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  /* statistics of the run */
  uint64_t n_passes = 0, n_replay = 0, n_records = 0, bytes_up = 0;
  double t_wait = 0, t_dev = 0;
};

namespace {
struct guard {
  int prev = -1;
  explicit guard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != dev) {
      (void)hipSetDevice(dev);
      prev = cur;
    }
  }
  ~guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct carver { /* arrays of a pass out of one allocation */
  char *base;
  size_t off = 0;
  explicit carver(void *p) : base((char *)p) {}
  template <class T> T *take(size_t n) {
    off = (off + 255u) & ~(size_t)255u;
    T *r = base ? (T *)(base + off) : nullptr;
    off += n * sizeof(T);
    return r;
  }
};

inline unsigned grid(uint64_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }
double now_s() {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

size_t prim_tmp_bytes(uint32_t n) {
  size_t m = 0, b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
  m = b > m ? b : m;
  (void)rocprim::exclusive_scan(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, n, rocprim::plus<unsigned long long>(),
                                (hipStream_t)0);
  m = b > m ? b : m;
  (void)rocprim::inclusive_scan(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, n, rocprim::maximum<unsigned long long>(),
                                (hipStream_t)0);
  m = b > m ? b : m;
  (void)rocprim::inclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
  m = b > m ? b : m;
  (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0,
                                  64, (hipStream_t)0);
  m = b > m ? b : m;
  return m + 256;
}

const char *replay_text(int e) {
  switch (e) {
    case BD_E_TID: return "a mapped record without a valid reference id";
    case BD_E_MATES_DISAGREE: return "the mates of a read disagree on their positions";
    case BD_E_DUP_NAME: return "duplicate read name";
    case BD_E_MATE_OPENS_BLOCK: return "input not sorted by coordinate (a mate opens a block)";
    case BD_E_BLOCK_START: return "a block whose first template starts right of the block's end (a mate position that is negative in the file, or input not sorted by coordinate)";
    case BD_E_NOSEQ_DUP: return "a duplicate of a template without bases";
    default: return "malformed record";
  }
}

/* slabs of the stream into the arena until the pass has enough new bytes (or the stream ends); parses the new records */
int pull_and_parse(bsc_bamdev *r, bool *final) {
  const double t0 = now_s();
  struct slab_rec {
    uint64_t stream_off;
    uint32_t n_recs;
    size_t rel_off; /* into recoff, dwords */
  };
  std::vector<slab_rec> got;
  size_t rel_total = 0;
  uint64_t new_bytes = 0;
  bsc_bam_slab prev;
  bool have_prev = false;
  int evi = 0;
  bam_cnt *cnt = (bam_cnt *)r->cnt.p;
  const uint32_t n_desc0 = r->n_desc;
  uint64_t last_rec_off = 0;
  bool any_rec = false;
  while (!r->stream_end && new_bytes < r->pass_bytes) {
    bsc_bam_slab sl;
    const double tw = now_s();
    const int rc = bsc_bamstream_next(r->bs, &sl);
    r->t_wait += now_s() - tw;
    if (rc < 0) return rc;
    if (rc == 0) {
      r->stream_end = true;
      break;
    }
    if (r->arena_end == 0 && r->arena.p == nullptr) r->arena_base = r->live_off = r->arena_end = sl.stream_off;
    /* room: the live bytes stay where they are unless the buffer must grow; then only they move */
    const uint64_t need = (r->arena_end - r->arena_base) + sl.n_bytes + 64;
    if (need > r->arena.cap) {
      const uint64_t live = r->arena_end - r->live_off;
      dev_buf nb;
      const size_t want = (size_t)(live + sl.n_bytes + r->pass_bytes + (64u << 20));
      int e = buf_reserve(nb, want);
      if (e) return e;
      if (live) {
        BAM_TRY(hipMemcpyAsync(nb.p, (const char *)r->arena.p + (r->live_off - r->arena_base), (size_t)live, hipMemcpyDeviceToDevice, r->s));
        BAM_TRY(hipStreamSynchronize(r->s));
      }
      buf_free(r->arena);
      r->arena = nb;
      r->arena_base = r->live_off;
    }
    BAM_TRY(hipMemcpyAsync((char *)r->arena.p + (sl.stream_off - r->arena_base), sl.bytes, sl.n_bytes, hipMemcpyHostToDevice, r->s));
    if (sl.n_recs) {
      int e = buf_reserve(r->recoff, (rel_total + sl.n_recs) * 4u, true, rel_total * 4u, r->s);
      if (e) return e;
      BAM_TRY(hipMemcpyAsync((uint32_t *)r->recoff.p + rel_total, sl.rec_off, (size_t)sl.n_recs * 4u, hipMemcpyHostToDevice, r->s));
      got.push_back({sl.stream_off, sl.n_recs, rel_total});
      rel_total += sl.n_recs;
      last_rec_off = sl.stream_off + sl.rec_off[sl.n_recs - 1];
      any_rec = true;
    }
    BAM_TRY(hipEventRecord(r->ev[evi], r->s));
    if (have_prev) { /* the previous slab's copy has the other event */
      BAM_TRY(hipEventSynchronize(r->ev[evi ^ 1]));
      int e = bsc_bamstream_release(r->bs, &prev);
      if (e) return e;
    }
    prev = sl;
    have_prev = true;
    evi ^= 1;
    r->arena_end = sl.stream_off + sl.n_bytes;
    new_bytes += sl.n_bytes;
    r->bytes_up += sl.n_bytes;
    if (sl.last) r->stream_end = true;
  }
  if (have_prev) {
    BAM_TRY(hipEventSynchronize(r->ev[evi ^ 1]));
    int e = bsc_bamstream_release(r->bs, &prev);
    if (e) return e;
  }
  *final = r->stream_end;
  /* descriptors: the pending record first, then the slabs' */
  const uint64_t n_new = rel_total + (r->has_pend ? 1u : 0u);
  if ((uint64_t)r->n_desc + n_new > 0x7ffffff0ull) return bsc_set_error(BSC_ERR_ARG, "device BAM reader: more than 2^31 records in one block");
  int e = buf_reserve(r->desc, ((size_t)r->n_desc + n_new + 1) * sizeof(bd_desc), true, (size_t)r->n_desc * sizeof(bd_desc), r->s);
  if (e) return e;
  bool pend_is_last = false;
  uint64_t pend_was = 0;
  if (r->has_pend) {
    hipLaunchKernelGGL(bsc_bam_parse_kernel, dim3(1), dim3(64), 0, r->s, (const uint8_t *)r->arena.p, r->arena_base, r->arena_end, r->pend_off,
                       (const uint32_t *)r->zero.p, 1u, r->par, (bd_desc *)r->desc.p, r->n_desc, cnt);
    r->n_desc++;
    pend_was = r->pend_off;
    pend_is_last = !any_rec;
    r->has_pend = false;
  }
  for (const slab_rec &g : got) {
    hipLaunchKernelGGL(bsc_bam_parse_kernel, dim3(grid(g.n_recs)), dim3(256), 0, r->s, (const uint8_t *)r->arena.p, r->arena_base, r->arena_end, g.stream_off,
                       (const uint32_t *)r->recoff.p + g.rel_off, g.n_recs, r->par, (bd_desc *)r->desc.p, r->n_desc, cnt);
    r->n_desc += g.n_recs;
  }
  BAM_TRY(hipGetLastError());
  /* the verdict on the new records: an incomplete LAST record waits for the next pass, anything else is an error */
  bam_cnt h;
  BAM_TRY(hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, r->s));
  BAM_TRY(hipStreamSynchronize(r->s));
  if (h.first_err != ~0ull) {
    bd_desc bad;
    BAM_TRY(hipMemcpy(&bad, (const bd_desc *)r->desc.p + h.first_err, sizeof bad, hipMemcpyDeviceToHost));
    if (bad.status == BD_ST_ERR_TRUNC && !*final && h.first_err == (unsigned long long)r->n_desc - 1ull) {
      r->has_pend = true;
      r->pend_off = any_rec ? last_rec_off : pend_was;
      (void)pend_is_last;
      r->n_desc--;
      const unsigned long long none = ~0ull;
      BAM_TRY(hipMemcpyAsync(&cnt->first_err, &none, sizeof none, hipMemcpyHostToDevice, r->s));
    } else if (bad.status == BD_ST_ERR_SIZE)
      return bsc_set_error(BSC_ERR_ARG, "BAM: implausible record size %u", bad.bs);
    else if (bad.status == BD_ST_ERR_TRUNC)
      return bsc_set_error(BSC_ERR_ARG, "BAM: input truncated");
    else
      return bsc_set_error(BSC_ERR_ARG, "BAM: malformed record");
  }
  r->n_records += r->n_desc - n_desc0;
  r->t_dev += now_s() - t0;
  return BSC_OK;
}

/* read_input over the descriptors in hand: the blocks that are complete come out (r->blocks), the rest is carried */
int segment_and_assemble(bsc_bamdev *r, bool final) {
  const double t0 = now_s();
  const uint32_t n = r->n_desc;
  r->blocks.clear();
  r->next_blk = 0;
  bam_cnt *cnt = (bam_cnt *)r->cnt.p;
  if (n == 0) return BSC_OK;
  hipStream_t s = r->s;
  /* the pass's arrays */
  uint32_t tab_size = 16;
  while (tab_size < 2u * n + 2u) tab_size *= 2u;
  const size_t n1 = (size_t)n + 1u;
  carver sizes(nullptr);
  for (int round = 0; round < 2; round++) {
    carver c(round ? r->ws_mem.p : nullptr);
    uint32_t *flag = c.take<uint32_t>(n1), *pos = c.take<uint32_t>(n1), *U = c.take<uint32_t>(n1);
    uint32_t *occ = c.take<uint32_t>(n1), *side0 = c.take<uint32_t>(n1), *side1 = c.take<uint32_t>(n1), *waiting = c.take<uint32_t>(n1),
             *ent_slot = c.take<uint32_t>(n1), *max_at = c.take<uint32_t>(n1), *slot_list = c.take<uint32_t>(n1);
    uint8_t *ent_alive = c.take<uint8_t>(n1), *slot_made = c.take<uint8_t>(n1), *blk_open = c.take<uint8_t>(n1);
    uint32_t *rf = c.take<uint32_t>(n1), *run = c.take<uint32_t>(n1), *flags = c.take<uint32_t>(n1), *f_blk = c.take<uint32_t>(n1), *f_grp = c.take<uint32_t>(n1),
             *blk_of = c.take<uint32_t>(n1), *grp_of = c.take<uint32_t>(n1), *g_first = c.take<uint32_t>(n1), *partner = c.take<uint32_t>(n1),
             *win0 = c.take<uint32_t>(n1), *win1 = c.take<uint32_t>(n1), *nval = c.take<uint32_t>(n1), *sval = c.take<uint32_t>(n1);
    unsigned long long *key = c.take<unsigned long long>(n1), *scan = c.take<unsigned long long>(n1), *nkey = c.take<unsigned long long>(n1),
                       *skey = c.take<unsigned long long>(n1);
    unsigned long long *len = c.take<unsigned long long>(n1), *nms = c.take<unsigned long long>(n1), *del = c.take<unsigned long long>(n1),
                       *seq_off = c.take<unsigned long long>(n1), *ms_off = c.take<unsigned long long>(n1), *del_off = c.take<unsigned long long>(n1);
    uint32_t *slot = c.take<uint32_t>(n1), *bop = c.take<uint32_t>(n1), *tidx = c.take<uint32_t>(n1), *bidx = c.take<uint32_t>(n1), *bstart = c.take<uint32_t>(n1),
             *tslot = c.take<uint32_t>(n1);
    uint32_t *tab = c.take<uint32_t>(tab_size), *tab_blk = c.take<uint32_t>(tab_size);
    if (!round) {
      int e = buf_reserve(r->ws_mem, c.off + 4096);
      if (e) return e;
      if ((e = buf_reserve(r->tmp, prim_tmp_bytes((uint32_t)n1)))) return e;
      continue;
    }
    void *tmp = r->tmp.p;
    size_t tb = r->tmp.cap;
    const bd_desc *D = (const bd_desc *)r->desc.p;
    /* the records that passed, in file order */
    hipLaunchKernelGGL(bsc_bam_flag_used_kernel, dim3(grid(n1)), dim3(256), 0, s, D, n, flag);
    BAM_TRY(rocprim::exclusive_scan(tmp, tb, flag, pos, 0u, n1, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(bsc_bam_scatter_used_kernel, dim3(grid(n)), dim3(256), 0, s, flag, pos, n, U);
    uint32_t n_used = 0;
    BAM_TRY(hipMemcpyAsync(&n_used, pos + n, 4, hipMemcpyDeviceToHost, s));
    BAM_TRY(hipStreamSynchronize(s));
    bd_ws ws;
    memset(&ws, 0, sizeof ws);
    ws.arena = (const uint8_t *)r->arena.p;
    ws.arena_base = r->arena_base;
    ws.D = D;
    ws.U = U;
    ws.n_used = n_used;
    ws.occ = occ;
    ws.side0 = side0;
    ws.side1 = side1;
    ws.waiting = waiting;
    ws.ent_slot = ent_slot;
    ws.ent_alive = ent_alive;
    ws.slot_made = slot_made;
    ws.blk_open = blk_open;
    ws.max_at = max_at;
    ws.slot_list = slot_list;
    ws.cts = cnt->cts;
    ws.bases = cnt->bases;
    ws.err = cnt->err;
    ws.tab = tab;
    ws.tab_blk = tab_blk;
    ws.tab_mask = tab_size - 1u;
    uint32_t n_done = 0;
    if (n_used == 0) { /* nothing passed: every record in hand is spent */
      r->n_desc = 0;
      r->live_off = r->has_pend ? r->pend_off : r->arena_end;
      return BSC_OK;
    }
    /* counters of the segmentation are committed only with the blocks they belong to: remember the state to fall back to */
    bam_cnt h0;
    BAM_TRY(hipMemcpyAsync(&h0, cnt, sizeof h0, hipMemcpyDeviceToHost, s));
    BAM_TRY(hipStreamSynchronize(s));
    bool decided = false;
    const bool force_replay = getenv("BSC_BAMDEV_REPLAY") != nullptr; /* tests: the slow path on ordinary input */
    if (!force_replay) {
      const unsigned g = grid(n_used);
      BAM_TRY(hipMemsetAsync(&cnt->irregular, 0, 4 * sizeof(uint32_t), s));
      hipLaunchKernelGGL(bsc_bam_run_flag_kernel, dim3(g), dim3(256), 0, s, ws, rf);
      BAM_TRY(rocprim::inclusive_scan(tmp, tb, rf, run, n_used, rocprim::plus<uint32_t>(), s));
      hipLaunchKernelGGL(bsc_bam_key_kernel, dim3(g), dim3(256), 0, s, ws, run, key, nkey, nval);
      BAM_TRY(rocprim::inclusive_scan(tmp, tb, key, scan, n_used, rocprim::maximum<unsigned long long>(), s));
      hipLaunchKernelGGL(bsc_bam_open_kernel, dim3(g), dim3(256), 0, s, ws, r->par, scan, flags, f_blk, f_grp, cnt);
      BAM_TRY(rocprim::inclusive_scan(tmp, tb, f_blk, blk_of, n_used, rocprim::plus<uint32_t>(), s));
      BAM_TRY(rocprim::inclusive_scan(tmp, tb, f_grp, grp_of, n_used, rocprim::plus<uint32_t>(), s));
      bam_cnt h1;
      BAM_TRY(hipMemcpyAsync(&h1, cnt, sizeof h1, hipMemcpyDeviceToHost, s));
      BAM_TRY(hipStreamSynchronize(s));
      const uint32_t u_done = final ? n_used : h1.u_done; /* the last block is complete only when the input has ended */
      if (!h1.irregular && u_done == 0) { /* one block in hand, and it goes on: nothing to hand out yet */
        r->t_dev += now_s() - t0;
        return BSC_OK;
      }
      if (!h1.irregular) {
        hipLaunchKernelGGL(bsc_bam_group_first_kernel, dim3(g), dim3(256), 0, s, flags, grp_of, n_used, g_first);
        BAM_TRY(hipMemsetAsync(partner, 0xff, (size_t)n_used * 4u, s));
        BAM_TRY(hipMemsetAsync(win0, 0, (size_t)n_used * 4u, s));
        BAM_TRY(hipMemsetAsync(win1, 0, (size_t)n_used * 4u, s));
        BAM_TRY(rocprim::radix_sort_pairs(tmp, tb, nkey, skey, nval, sval, n_used, 0, 64, s));
        hipLaunchKernelGGL(bsc_bam_chain_kernel, dim3(g), dim3(256), 0, s, ws, skey, sval, blk_of, partner, cnt);
        hipLaunchKernelGGL(bsc_bam_group_kernel, dim3(g), dim3(256), 0, s, ws, r->par, g_first, grp_of, u_done, cnt);
        hipLaunchKernelGGL(bsc_bam_join_kernel, dim3(g), dim3(256), 0, s, ws, r->par, partner, blk_of, scan, win0, win1, u_done, cnt);
        hipLaunchKernelGGL(bsc_bam_wins_kernel, dim3(g), dim3(256), 0, s, ws, win0, win1, u_done);
        BAM_TRY(hipGetLastError());
        BAM_TRY(hipMemcpyAsync(&h1, cnt, sizeof h1, hipMemcpyDeviceToHost, s));
        BAM_TRY(hipStreamSynchronize(s));
      }
      if (!h1.irregular) {
        decided = true;
        n_done = u_done;
      } else { /* the counters as they were: the replay counts again */
        BAM_TRY(hipMemcpyAsync(cnt->cts, h0.cts, sizeof h0.cts, hipMemcpyHostToDevice, s));
        BAM_TRY(hipMemcpyAsync(cnt->bases, h0.bases, sizeof h0.bases, hipMemcpyHostToDevice, s));
      }
    }
    if (!decided) {
      BAM_TRY(hipMemsetAsync(tab, 0, (size_t)tab_size * 4u, s));
      BAM_TRY(hipMemsetAsync(tab_blk, 0, (size_t)tab_size * 4u, s));
      hipLaunchKernelGGL(bsc_bam_replay_kernel, dim3(1), dim3(64), 0, s, ws, r->par, final ? 1 : 0, cnt);
      BAM_TRY(hipGetLastError());
      bam_cnt h2;
      BAM_TRY(hipMemcpyAsync(&h2, cnt, sizeof h2, hipMemcpyDeviceToHost, s));
      BAM_TRY(hipStreamSynchronize(s));
      r->n_replay++;
      if (h2.err[0]) return bsc_set_error(BSC_ERR_ARG, "BAM: %s (record %llu of the records in hand that passed the filters)", replay_text((int)h2.err[0]), h2.err[1]);
      n_done = h2.n_done;
    }
    if (n_done == 0) {
      r->t_dev += now_s() - t0;
      return BSC_OK;
    }
    /* assembly of the complete blocks */
    const unsigned gd = grid((uint64_t)n_done + 1u);
    hipLaunchKernelGGL(bsc_bam_asm_in_kernel, dim3(gd), dim3(256), 0, s, ws, n_done, len, nms, del, slot, bop);
    BAM_TRY(rocprim::exclusive_scan(tmp, tb, len, seq_off, 0ull, n_done + 1u, rocprim::plus<unsigned long long>(), s));
    BAM_TRY(rocprim::exclusive_scan(tmp, tb, nms, ms_off, 0ull, n_done + 1u, rocprim::plus<unsigned long long>(), s));
    BAM_TRY(rocprim::exclusive_scan(tmp, tb, del, del_off, 0ull, n_done + 1u, rocprim::plus<unsigned long long>(), s));
    BAM_TRY(rocprim::exclusive_scan(tmp, tb, slot, tidx, 0u, n_done + 1u, rocprim::plus<uint32_t>(), s));
    BAM_TRY(rocprim::exclusive_scan(tmp, tb, bop, bidx, 0u, n_done + 1u, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(bsc_bam_block_first_kernel, dim3(gd), dim3(256), 0, s, bop, bidx, slot, tidx, n_done, bstart, tslot);
    struct {
      unsigned long long seq, ms;
      uint32_t n_tpl, n_blk;
    } tot;
    BAM_TRY(hipMemcpyAsync(&tot.seq, seq_off + n_done, 8, hipMemcpyDeviceToHost, s));
    BAM_TRY(hipMemcpyAsync(&tot.ms, ms_off + n_done, 8, hipMemcpyDeviceToHost, s));
    BAM_TRY(hipMemcpyAsync(&tot.n_tpl, tidx + n_done, 4, hipMemcpyDeviceToHost, s));
    BAM_TRY(hipMemcpyAsync(&tot.n_blk, bidx + n_done, 4, hipMemcpyDeviceToHost, s));
    BAM_TRY(hipStreamSynchronize(s));
    int e;
    if ((e = buf_reserve(r->d_tpl, ((size_t)tot.n_tpl + 1) * sizeof(bd_raw_template)))) return e;
    if ((e = buf_reserve(r->d_seq, (size_t)tot.seq + 64))) return e;
    if ((e = buf_reserve(r->d_ms, ((size_t)tot.ms + 1) * sizeof(bd_misms)))) return e;
    if ((e = buf_reserve(r->d_tab, ((size_t)tot.n_blk + 1) * sizeof(bam_blk)))) return e;
    const unsigned none = 0xffffffffu;
    BAM_TRY(hipMemcpyAsync(&cnt->blk_err, &none, 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(bsc_bam_block_table_kernel, dim3(grid(tot.n_blk)), dim3(256), 0, s, ws, bstart, tot.n_blk, n_done, tidx, tslot, seq_off, ms_off, del_off,
                       (bam_blk *)r->d_tab.p, cnt);
    hipLaunchKernelGGL(bsc_bam_template_kernel, dim3(grid(n_done)), dim3(256), 0, s, ws, n_done, slot, tidx, bidx, (const bam_blk *)r->d_tab.p, seq_off, ms_off,
                       (bd_raw_template *)r->d_tpl.p);
    hipLaunchKernelGGL(bsc_bam_misms_kernel, dim3(grid(n_done)), dim3(256), 0, s, ws, n_done, ms_off, (bd_misms *)r->d_ms.p);
    {
      unsigned gw = (n_done + BAM_DEC_WAVES - 1u) / BAM_DEC_WAVES;
      if (gw > 65536u) gw = 65536u;
      hipLaunchKernelGGL(bsc_bam_decode_kernel, dim3(gw), dim3(64 * BAM_DEC_WAVES), 0, s, ws, n_done, seq_off, (uint8_t *)r->d_seq.p);
    }
    BAM_TRY(hipGetLastError());
    std::vector<bam_blk> tabh(tot.n_blk);
    bam_cnt h3;
    if (tot.n_blk) BAM_TRY(hipMemcpyAsync(tabh.data(), r->d_tab.p, (size_t)tot.n_blk * sizeof(bam_blk), hipMemcpyDeviceToHost, s));
    BAM_TRY(hipMemcpyAsync(&h3, cnt, sizeof h3, hipMemcpyDeviceToHost, s));
    /* what is carried: the records from the block in hand on */
    uint32_t carry_rec = n; /* record index */
    if (n_done < n_used) BAM_TRY(hipMemcpyAsync(&carry_rec, U + n_done, 4, hipMemcpyDeviceToHost, s));
    BAM_TRY(hipStreamSynchronize(s));
    if (h3.blk_err != 0xffffffffu) return bsc_set_error(BSC_ERR_ARG, "BAM: %s", replay_text(BD_E_BLOCK_START));
    for (const bam_blk &b : tabh)
      if (b.n_tpl) r->blocks.push_back(b);
    /* the carried descriptors move to the front (through the second buffer); the bytes before the first of them are dead */
    uint64_t live = r->has_pend ? r->pend_off : r->arena_end;
    const uint32_t n_carry = n - carry_rec;
    if (n_carry) {
      bd_desc first;
      BAM_TRY(hipMemcpy(&first, D + carry_rec, sizeof first, hipMemcpyDeviceToHost));
      live = first.off;
      if ((e = buf_reserve(r->desc2, ((size_t)n_carry + 1) * sizeof(bd_desc)))) return e;
      BAM_TRY(hipMemcpyAsync(r->desc2.p, D + carry_rec, (size_t)n_carry * sizeof(bd_desc), hipMemcpyDeviceToDevice, s));
      BAM_TRY(hipStreamSynchronize(s));
      dev_buf t = r->desc;
      r->desc = r->desc2;
      r->desc2 = t;
    }
    r->n_desc = n_carry;
    r->live_off = live;
  }
  r->t_dev += now_s() - t0;
  return BSC_OK;
}
} // namespace

/* ---- C ABI ---------------------------------------------------------------------------------------------------------------------- */
extern "C" void bsc_bamdev_close(bsc_bamdev *r) {
  if (!r) return;
  {
    guard g(r->device);
    if (r->s) (void)hipStreamSynchronize(r->s);
    if (r->bs) bsc_bamstream_close(r->bs);
    dev_buf *all[] = {&r->arena, &r->desc, &r->desc2, &r->recoff, &r->cnt, &r->zero, &r->ws_mem, &r->tmp, &r->d_tpl, &r->d_seq, &r->d_ms, &r->d_tab, &r->d_keep};
    for (dev_buf *b : all) buf_free(*b);
    for (int i = 0; i < 2; i++)
      if (r->ev[i]) (void)hipEventDestroy(r->ev[i]);
  }
  delete r;
}

extern "C" int bsc_bamdev_open_contigs(bsc_context *ctx, const char *path, int n_threads, const int32_t *tids, int n_tids, bsc_bamdev **out);
extern "C" int bsc_bamdev_open(bsc_context *ctx, const char *path, int n_threads, bsc_bamdev **out) {
  return bsc_bamdev_open_contigs(ctx, path, n_threads, nullptr, -1, out);
}

extern "C" int bsc_bamdev_open_contigs(bsc_context *ctx, const char *path, int n_threads, const int32_t *tids, int n_tids, bsc_bamdev **out) {
  if (!ctx || !path || !out || (n_tids > 0 && !tids)) return bsc_set_error(BSC_ERR_ARG, "bsc_bamdev_open: NULL argument");
  *out = nullptr;
  bsc_bamdev *r = new (std::nothrow) bsc_bamdev;
  if (!r) return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamdev_open: out of memory");
  r->ctx = ctx;
  r->device = bsc_ctx_device(ctx);
  r->s = (hipStream_t)bsc_ctx_stream(ctx);
  guard g(r->device);
  const char *pb = getenv("BSC_BAMDEV_PASS_MB"), *pk = getenv("BSC_BAMDEV_PASS_KB"); /* new bytes per device pass (tests: small passes) */
  if (pb && atoi(pb) > 0) r->pass_bytes = (uint64_t)atoi(pb) << 20;
  if (pk && atoi(pk) > 0) r->pass_bytes = (uint64_t)atoi(pk) << 10;
  const char *sb = getenv("BSC_BAMDEV_SLAB_KB"); /* tests: small slabs, so that records straddle them */
  int rc = bsc_bamstream_open_contigs(path, n_threads, sb && atoi(sb) > 0 ? (uint64_t)atoi(sb) << 10 : 0, 0, tids, n_tids, &r->bs);
  if (rc) {
    bsc_bamdev_close(r);
    return rc;
  }
  if (n_tids >= 0) { /* the selection as the record parser's filter */
    const int n_ref = bsc_bamstream_n_refs(r->bs);
    std::vector<uint8_t> keep((size_t)n_ref + 1, 0);
    for (int i = 0; i < n_tids; i++) {
      if (tids[i] >= n_ref) {
        bsc_bamdev_close(r);
        return bsc_set_error(BSC_ERR_ARG, "bsc_bamdev_open_contigs: contig %d is not in the header (%d contigs)", tids[i], n_ref);
      }
      if (tids[i] < 0) r->keep_unplaced = true;
      else keep[(size_t)tids[i]] = 1;
    }
    if ((rc = buf_reserve(r->d_keep, keep.size())) || hipMemcpy(r->d_keep.p, keep.data(), keep.size(), hipMemcpyHostToDevice) != hipSuccess) {
      bsc_bamdev_close(r);
      return rc ? rc : bsc_set_error(BSC_ERR_HIP, "bsc_bamdev_open_contigs: device set-up failed");
    }
  }
  if ((rc = buf_reserve(r->cnt, sizeof(bam_cnt))) || (rc = buf_reserve(r->zero, 64))) {
    bsc_bamdev_close(r);
    return rc;
  }
  bam_cnt h;
  memset(&h, 0, sizeof h);
  h.first_err = ~0ull;
  if (hipMemcpy(r->cnt.p, &h, sizeof h, hipMemcpyHostToDevice) != hipSuccess || hipMemset(r->zero.p, 0, 64) != hipSuccess ||
      hipStreamSynchronize(nullptr) != hipSuccess || /* (the memset is done before the reader's stream, which does not wait for the null stream, reads it) */
      hipEventCreateWithFlags(&r->ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->ev[1], hipEventDisableTiming) != hipSuccess) {
    bsc_bamdev_close(r);
    return bsc_set_error(BSC_ERR_HIP, "bsc_bamdev_open: device set-up failed");
  }
  *out = r;
  return BSC_OK;
}

extern "C" int bsc_bamdev_n_refs(const bsc_bamdev *r) { return r ? bsc_bamstream_n_refs(r->bs) : 0; }
extern "C" const char *bsc_bamdev_ref_name(const bsc_bamdev *r, int i) { return r ? bsc_bamstream_ref_name(r->bs, i) : nullptr; }
extern "C" uint32_t bsc_bamdev_ref_len(const bsc_bamdev *r, int i) { return r ? bsc_bamstream_ref_len(r->bs, i) : 0; }
extern "C" const char *bsc_bamdev_header_text(const bsc_bamdev *r) { return r ? bsc_bamstream_header_text(r->bs) : nullptr; }

extern "C" int bsc_bamdev_next_block(bsc_bamdev *r, const bsc_reader_params *par, bsc_dev_read_block *blk) {
  if (!r || !par || !blk) return bsc_set_error(BSC_ERR_ARG, "bsc_bamdev_next_block: NULL argument");
  memset(blk, 0, sizeof *blk);
  guard g(r->device);
  bd_params p;
  memset(&p, 0, sizeof p);
  p.mapq_thresh = par->mapq_thresh;
  p.max_template_len = par->max_template_len;
  p.keep_unmatched = par->keep_unmatched != 0;
  p.ignore_duplicates = par->ignore_duplicates != 0;
  p.keep_duplicates = par->keep_duplicates != 0;
  p.region_tid = par->region_tid;
  p.region_start = par->region_start;
  p.region_stop = par->region_stop;
  p.n_ref = bsc_bamstream_n_refs(r->bs);
  p.tid_keep = (const uint8_t *)r->d_keep.p;
  p.keep_unplaced = r->keep_unplaced ? 1u : 0u;
  if (r->par_set && memcmp(&p, &r->par, sizeof p)) return bsc_set_error(BSC_ERR_ARG, "bsc_bamdev_next_block: the reader's parameters changed in mid-file");
  r->par = p;
  r->par_set = true;
  while (r->next_blk >= r->blocks.size()) {
    if (r->finished) return 0;
    bool final = false;
    int rc = pull_and_parse(r, &final);
    if (rc) return rc;
    if ((rc = segment_and_assemble(r, final))) return rc;
    r->n_passes++;
    if (final) {
      r->finished = true;
      bam_cnt h;
      BAM_TRY(hipMemcpy(&h, r->cnt.p, sizeof h, hipMemcpyDeviceToHost));
      memcpy(r->cts, h.cts, sizeof r->cts);
      memcpy(r->bases, h.bases, sizeof r->bases);
      r->malformed = h.malformed;
    }
  }
  const bam_blk &b = r->blocks[r->next_blk++];
  blk->tid = b.tid;
  blk->y = b.y;
  blk->x = b.x0 > 2 ? b.x0 - 2 : 1; /* bsc_block_start */
  blk->nr = b.n_tpl;
  blk->d_tpl = (const char *)r->d_tpl.p + (size_t)b.first_tpl * sizeof(bd_raw_template);
  blk->d_seq = (const char *)r->d_seq.p + b.seq0;
  blk->seq_bytes = b.seq_bytes;
  blk->d_misms = (const char *)r->d_ms.p + (size_t)b.ms0 * sizeof(bd_misms);
  blk->n_misms = b.n_ms;
  blk->ins_pad = b.ins_pad;
  return 1;
}

extern "C" int bsc_bamdev_filter_counts(bsc_bamdev *r, uint64_t cts[15], uint64_t bases[15]) {
  if (!r) return bsc_set_error(BSC_ERR_ARG, "bsc_bamdev_filter_counts: NULL argument");
  guard g(r->device);
  bam_cnt h;
  BAM_TRY(hipStreamSynchronize(r->s));
  BAM_TRY(hipMemcpy(&h, r->cnt.p, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < 15; i++) {
    if (cts) cts[i] = h.cts[i];
    if (bases) bases[i] = h.bases[i];
  }
  r->malformed = h.malformed;
  return BSC_OK;
}
extern "C" uint64_t bsc_bamdev_malformed(bsc_bamdev *r) {
  if (!r) return 0;
  (void)bsc_bamdev_filter_counts(r, nullptr, nullptr);
  return r->malformed;
}

/* a block's arrays to the host (tests, and callers that want bsc_read_block's view): tpl[nr], seq[seq_bytes], misms[n_misms] */
extern "C" int bsc_bamdev_fetch_block(bsc_bamdev *r, const bsc_dev_read_block *blk, bsc_raw_template *tpl, uint8_t *seq, bsc_misms *misms) {
  if (!r || !blk) return bsc_set_error(BSC_ERR_ARG, "bsc_bamdev_fetch_block: NULL argument");
  guard g(r->device);
  BAM_TRY(hipStreamSynchronize(r->s));
  if (tpl && blk->nr) BAM_TRY(hipMemcpy(tpl, blk->d_tpl, (size_t)blk->nr * sizeof *tpl, hipMemcpyDeviceToHost));
  if (seq && blk->seq_bytes) BAM_TRY(hipMemcpy(seq, blk->d_seq, (size_t)blk->seq_bytes, hipMemcpyDeviceToHost));
  if (misms && blk->n_misms) BAM_TRY(hipMemcpy(misms, blk->d_misms, (size_t)blk->n_misms * sizeof *misms, hipMemcpyDeviceToHost));
  return BSC_OK;
}

/* how the run went: {passes, passes the replay decided, records parsed, bytes uploaded}, seconds waiting for slabs, seconds in device passes */
extern "C" void bsc_bamdev_run_stats(const bsc_bamdev *r, uint64_t counts[4], double seconds[2]) {
  if (!r) return;
  if (counts) {
    counts[0] = r->n_passes;
    counts[1] = r->n_replay;
    counts[2] = r->n_records;
    counts[3] = r->bytes_up;
  }
  if (seconds) {
    seconds[0] = r->t_wait;
    seconds[1] = r->t_dev;
  }
}
