// TEST INFRASTRUCTURE — csrc/bamdev_core.h (the statements of the device BAM reader) run by plain loops on the CPU, so that the
// reader's LOGIC can be checked against csrc/bamio.c and oracle/py_bam.py in a container without a GPU.  The product never loads
// this: the kernels of csrc/bamdev.hip call the same functions per lane, with rocPRIM's scans and sort where this file loops.
// Built by tests/test_bamdev_emul.py with g++.  The pass structure mirrors csrc/bamdev.hip's bsc_dev_bam_pass.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../bs_call_amd/csrc/bamdev_core.h"

namespace {
struct Block {
  int32_t tid;
  uint32_t y;
  std::vector<bd_raw_template> tpl;
  std::vector<uint8_t> seq;
  std::vector<bd_misms> ms;
};
struct State {
  std::vector<Block> blocks;
  unsigned long long cts[15], bases[15], malformed;
  int err_code;
  char err_text[256];
  int used_replay, used_fast; // passes decided by each path
} G;

struct Pass {
  const uint8_t *arena;
  uint64_t arena_base, arena_len;
  bd_params par;
  std::vector<bd_desc> D;      // descriptors of the carried + new records
  std::vector<uint32_t> U;
};

int fail(int code, const char *msg) {
  G.err_code = code;
  snprintf(G.err_text, sizeof G.err_text, "%s", msg);
  return -1;
}

// one pass over the records D[0..): returns the number of RECORDS consumed (those of complete blocks and everything unused before
// the first record of the block in hand), or -1
long run_pass(Pass &P, int final, int mode) {
  const uint32_t n_rec = (uint32_t)P.D.size();
  P.U.clear();
  for (uint32_t r = 0; r < n_rec; r++)
    if (P.D[r].status == BD_ST_USE) P.U.push_back(r);
  const uint32_t n = (uint32_t)P.U.size();
  std::vector<uint32_t> occ(n + 1), side0(n + 1), side1(n + 1), waiting(n + 1), ent_slot(n + 1), max_at(n + 1), slot_list(n + 1);
  std::vector<uint8_t> ent_alive(n + 1), slot_made(n + 1), blk_open(n + 1);
  uint32_t tab_size = 16;
  while (tab_size < 2 * n + 2) tab_size *= 2;
  std::vector<uint32_t> tab(tab_size, 0), tab_blk(tab_size, 0);
  unsigned long long cts[15] = {0}, bases[15] = {0}, err[2] = {0, 0};
  bd_ws ws;
  memset(&ws, 0, sizeof ws);
  ws.arena = P.arena;
  ws.arena_base = P.arena_base;
  ws.D = P.D.data();
  ws.U = P.U.data();
  ws.n_used = n;
  ws.occ = occ.data();
  ws.side0 = side0.data();
  ws.side1 = side1.data();
  ws.waiting = waiting.data();
  ws.ent_slot = ent_slot.data();
  ws.ent_alive = ent_alive.data();
  ws.slot_made = slot_made.data();
  ws.blk_open = blk_open.data();
  ws.max_at = max_at.data();
  ws.slot_list = slot_list.data();
  ws.cts = cts;
  ws.bases = bases;
  ws.err = err;
  ws.tab = tab.data();
  ws.tab_blk = tab_blk.data();
  ws.tab_mask = tab_size - 1;

  uint32_t n_done = 0;
  bool decided = false;
  if (mode != 1 && n) { // the parallel path
    uint32_t irregular = 0;
    std::vector<uint64_t> scan(n);
    std::vector<uint32_t> run(n), flags(n), blk_of(n), partner(n, BD_NONE), win0(n, 0), win1(n, 0);
    uint32_t rn = 0;
    for (uint32_t u = 0; u < n; u++) {
      if (bd_f_run_start(ws, u)) rn++;
      run[u] = rn;
      const uint64_t k = bd_f_key(rn, BD_D(ws, u));
      scan[u] = u ? std::max(scan[u - 1], k) : k;
    }
    uint32_t b = 0;
    for (uint32_t u = 0; u < n; u++) {
      flags[u] = bd_f_open(ws, P.par, u, u ? (uint32_t)scan[u - 1] : 0u, &irregular);
      if (flags[u] & 1u) b++;
      blk_of[u] = b;
      blk_open[u] = (uint8_t)(flags[u] & 1u);
      max_at[u] = (uint32_t)scan[u];
    }
    uint32_t u_done = n;
    if (!final) {
      u_done = n;
      while (u_done > 0 && !(flags[u_done - 1] & 1u)) u_done--;
      u_done = u_done ? u_done - 1 : 0; // first record of the last block
    }
    // names: the paired records sorted by (run, hash), file order kept
    std::vector<uint32_t> sorted;
    std::vector<uint64_t> key(n);
    for (uint32_t u = 0; u < n; u++)
      if (BD_D(ws, u).aflag & BD_F_PAIRED) {
        key[u] = BD_D(ws, u).hash ^ ((uint64_t)run[u] * 0x9e3779b97f4a7c15ull);
        sorted.push_back(u);
      }
    std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t c) { return key[a] < key[c]; });
    for (size_t i = 0; i < sorted.size();) {
      size_t j = i + 1;
      while (j < sorted.size() && key[sorted[j]] == key[sorted[i]]) j++;
      bd_f_chain(ws, sorted.data(), (uint32_t)i, (uint32_t)j, blk_of.data(), partner.data(), &irregular);
      i = j;
    }
    for (uint32_t u = 0; u < u_done; u++) {
      slot_made[u] = 0;
      ent_alive[u] = 0;
    }
    for (uint32_t g = 0; g < u_done;) {
      uint32_t e = g + 1;
      while (e < u_done && !(flags[e] & 2u)) e++;
      bd_f_group(ws, P.par, g, e, 0, &irregular);
      g = e;
    }
    for (uint32_t u = 0; u < u_done; u++) bd_f_join(ws, P.par, u, partner.data(), blk_of.data(), scan.data(), win0.data(), win1.data(), &irregular);
    for (uint32_t s = 0; s < u_done; s++) {
      if (win0[s]) side0[s] = win0[s] - 1;
      if (win1[s]) side1[s] = win1[s] - 1;
    }
    if (!irregular) {
      decided = true;
      n_done = u_done;
      G.used_fast++;
    } else if (mode == 2)
      return fail(100, "irregular input: the parallel path does not decide it");
    else { // start over: the replay decides
      for (int i = 0; i < 15; i++) cts[i] = bases[i] = 0;
    }
  }
  if (!decided) {
    const int e = bd_replay(ws, P.par, 0, final, &n_done);
    if (e) {
      char msg[128];
      snprintf(msg, sizeof msg, "reader error %d at used record %llu", e, err[1]);
      return fail(e, msg);
    }
    if (n) G.used_replay++;
  }
  // decoded reads and lists of the used records of complete blocks, then the blocks
  std::vector<uint64_t> seq_off(n + 1, 0), ms_off(n + 1, 0);
  for (uint32_t u = 0; u < n_done; u++) {
    seq_off[u + 1] = seq_off[u] + BD_D(ws, u).l_seq;
    ms_off[u + 1] = ms_off[u] + BD_D(ws, u).n_ms;
  }
  for (uint32_t u = 0; u < n_done;) {
    uint32_t e = u + 1;
    while (e < n_done && !blk_open[e]) e++;
    Block B;
    B.tid = BD_D(ws, u).tid;
    B.y = max_at[e - 1];
    for (uint32_t s = u; s < e; s++)
      if (slot_made[s]) {
        bd_raw_template t;
        bd_template(ws, s, seq_off.data(), ms_off.data(), seq_off[u], ms_off[u], &t);
        B.tpl.push_back(t);
      }
    if (!B.tpl.empty()) {
      const uint32_t x0 = B.tpl[0].pos[0] ? B.tpl[0].pos[0] : B.tpl[0].pos[1];
      if (x0 == 0 || x0 > B.y) return fail(BD_E_BLOCK_START, "a block whose first template starts right of its end");
      B.seq.resize(seq_off[e] - seq_off[u]);
      B.ms.resize(ms_off[e] - ms_off[u]);
      for (uint32_t v = u; v < e; v++) {
        const bd_desc &d = BD_D(ws, v);
        const uint8_t *rec = P.arena + (d.off - P.arena_base);
        const uint8_t *seq4 = rec + 36 + d.l_name + 4 * d.n_cigar, *qual = seq4 + (d.l_seq + 1) / 2;
        for (uint32_t i = 0; i < d.l_seq; i++) B.seq[seq_off[v] - seq_off[u] + i] = bd_base_byte(seq4, qual, i);
        if (d.n_ms) bd_misms_of(rec, d, B.ms.data() + (ms_off[v] - ms_off[u]));
      }
      G.blocks.push_back(std::move(B));
    }
    u = e;
  }
  for (int i = 0; i < 15; i++) {
    G.cts[i] += cts[i];
    G.bases[i] += bases[i];
  }
  return n_done < n ? (long)P.U[n_done] : (long)n_rec;
}
} // namespace

extern "C" {
// stream[0 .. len): the inflated BAM file; rec_off[n_recs]: the records' offsets (the host walk's).  pass_recs: records per pass (0 = all
// in one), to exercise the carry of a block that a pass ends in.  mode 0: parallel path, the replay where it is irregular; 1: replay only;
// 2: parallel path only (irregular input is an error).  Returns 0 or -1 (bd_emul_error).
int bd_emul_run(const uint8_t *stream, uint64_t len, const uint64_t *rec_off, uint64_t n_recs, const bd_params *par, uint32_t pass_recs, int mode) {
  G.blocks.clear();
  memset(G.cts, 0, sizeof G.cts);
  memset(G.bases, 0, sizeof G.bases);
  G.malformed = 0;
  G.err_code = 0;
  G.err_text[0] = 0;
  G.used_replay = G.used_fast = 0;
  Pass P;
  P.arena = stream;
  P.arena_base = 0;
  P.arena_len = len;
  P.par = *par;
  uint64_t next = 0; // next record to parse
  if (pass_recs == 0) pass_recs = 0xffffffffu;
  for (;;) {
    const uint64_t take = std::min<uint64_t>(pass_recs, n_recs - next);
    for (uint64_t i = 0; i < take; i++) {
      bd_desc d;
      const uint64_t o = rec_off[next + i];
      if (o >= len) return fail(BD_E_RECORD, "record offset beyond the stream");
      bd_parse(stream + o, len - o, o, P.par, d);
      if (d.status >= BD_ST_ERR_SIZE) return fail(BD_E_RECORD, "malformed record");
      if (d.status == BD_ST_FILTERED) {
        G.cts[d.flt]++;
        G.bases[d.flt] += d.l_seq;
      } else if (d.status == BD_ST_MALFORMED_CIGAR)
        G.malformed++;
      P.D.push_back(d);
    }
    next += take;
    const int final = next == n_recs;
    const long used = run_pass(P, final, mode);
    if (used < 0) return -1;
    P.D.erase(P.D.begin(), P.D.begin() + used);
    if (final) break;
  }
  return 0;
}
const char *bd_emul_error(void) { return G.err_text; }
int bd_emul_error_code(void) { return G.err_code; }
uint64_t bd_emul_n_blocks(void) { return G.blocks.size(); }
void bd_emul_block(uint64_t i, int32_t *tid, uint32_t *y, uint32_t *nr, uint64_t *seq_bytes, uint64_t *n_ms) {
  const Block &b = G.blocks[i];
  *tid = b.tid;
  *y = b.y;
  *nr = (uint32_t)b.tpl.size();
  *seq_bytes = b.seq.size();
  *n_ms = b.ms.size();
}
void bd_emul_block_data(uint64_t i, void *tpl, void *seq, void *ms) {
  const Block &b = G.blocks[i];
  memcpy(tpl, b.tpl.data(), b.tpl.size() * sizeof(bd_raw_template));
  if (!b.seq.empty()) memcpy(seq, b.seq.data(), b.seq.size());
  if (!b.ms.empty()) memcpy(ms, b.ms.data(), b.ms.size() * sizeof(bd_misms));
}
void bd_emul_counts(uint64_t *cts, uint64_t *bases, uint64_t *malformed, int *used_replay, int *used_fast) {
  for (int i = 0; i < 15; i++) {
    cts[i] = G.cts[i];
    bases[i] = G.bases[i];
  }
  *malformed = G.malformed;
  *used_replay = G.used_replay;
  *used_fast = G.used_fast;
}
}
