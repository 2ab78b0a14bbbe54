"""Walking whole contigs through the fused chain in fixed windows, pile-ups resident in HBM — the host logic of
BASELINE.json configs[2] / [4] (a human-scale genome, contigs sharded across the GPUs of a node).

The reference's unit of parallelism is the contig: it is run one process per contig and the outputs concatenated
(reference README.md:73-76, src/process_sam_header.c:67-70), and inside a contig a block (one call_genotypes_ML) is a
maximal run of overlapping templates.  Here a rank owns whole contigs (longest-processing-time assignment, shard.py),
keeps their pile-ups in HBM and calls them in windows of 4 Mi positions through
bsc_chain_device, which hands each window the 2 positions / 4 reference bases of context the printer's sliding window
looks at — so the windows of a contig give exactly the records of the contig called in one piece.

Host logic only: nothing here computes calls; torch supplies the device buffers.
"""
from dataclasses import dataclass
from typing import List, Sequence

from . import shard

SEED = 88172645463325252
# Positions per window: SURVEY.md 8(d)'s 4 Mi, rounded down to a multiple of the fused kernel's 60-position wave-tile
# (69 905 x 60 = 4 194 300) so that no window but a contig's last ends in a partial tile (one launch less per window).
WINDOW = (shard.WINDOW // 60) * 60
PILEUP_BYTES, CORE_BYTES = 104, 64
BYTES_PER_POSITION = PILEUP_BYTES + 1 + CORE_BYTES  # resident: pile-up + reference code + record


def window_for(caller, limit: int = shard.WINDOW) -> int:
    """The largest window <= `limit` positions (4 Mi) in which every resident wave of the fused kernel runs the
    same number of tiles and nothing is left for a guarded launch (`bsc_chain_window_size`: 256 CUs x 16 waves x (60 + 62 k) on
    an MI355X -> k = 15, 4 055 040 positions: the first tile of a wave's run forms 60 records, every further one 62)."""
    return caller.window_size(limit)


def contig_first_sites(lengths: Sequence[int], pad: int = 64) -> List[int]:
    """Where each contig starts on the synthetic genome of the generators (contigs laid end to end, `pad` sites apart)."""
    out, s = [], 1
    for n in lengths:
        out.append(s)
        s += int(n) + pad
    return out


def rank_contigs(lengths: Sequence[int], rank: int, world_size: int) -> List[int]:
    """The contigs rank `rank` of `world_size` owns (LPT), in contig order."""
    return sorted(shard.assign_contigs(lengths, world_size)[rank])


def batches(contigs: Sequence[int], lengths: Sequence[int], budget_bytes: int) -> List[List[int]]:
    """Groups of whole contigs whose resident buffers fit `budget_bytes` (a contig larger than the budget is a group of
    its own: the caller then fails on allocation, loudly)."""
    out: List[List[int]] = []
    cur: List[int] = []
    used = 0
    for c in contigs:
        need = int(lengths[c]) * BYTES_PER_POSITION
        if cur and used + need > budget_bytes:
            out.append(cur)
            cur, used = [], 0
        cur.append(c)
        used += need
    if cur:
        out.append(cur)
    return out


@dataclass
class ResidentContig:
    """One contig's inputs and outputs in HBM (torch uint8 tensors)."""

    index: int
    length: int
    d_cts: object  # (length + 2) x 104 bytes: pile-ups (two spare records behind the contig)
    d_ref: object  # length + 2 reference codes
    d_core: object  # length x 64 bytes: bsc_vcf_core records
    d_dbsnp: object = None  # length rs_found flags, or None
    d_gc: object = None  # the contig's GC bins (ctg_stats->gc), or None
    gc_start: int = 0  # first A/C/G/T position of the contig


def make_resident(caller, index: int, length: int, first_site: int, coverage: int, device, flags: int = 1,
                  seed: int = SEED + 3, stream=None, dbsnp_flags=None, with_gc: bool = False):
    """Generate contig `index` on the device (the L-pileup generator; flags = 1: 1 % of 10-kb runs are N without reads).
    dbsnp_flags: uint8[length] rs_found per position (DbSnpIndex.flags(1, length)) for BASELINE.json configs[4].
    with_gc: also the contig's GC bins (what load_sequence computes when a report is asked for), for the report's
    GC-by-coverage table."""
    import torch

    d_cts = torch.empty((length + 2) * PILEUP_BYTES, dtype=torch.uint8, device=device)
    d_ref = torch.empty(length + 2, dtype=torch.uint8, device=device)
    d_core = torch.empty(length * CORE_BYTES, dtype=torch.uint8, device=device)
    caller.synth_device(seed, first_site, length + 2, coverage, d_cts.data_ptr(), d_ref.data_ptr(), flags, stream)
    d_db = None if dbsnp_flags is None else torch.from_numpy(dbsnp_flags).to(device)
    d_gc, gc_start = None, 0
    if with_gc:
        from .caller import gc_bins

        torch.cuda.synchronize()
        gc_start, bins = gc_bins(d_ref[:length].cpu().numpy())
        d_gc = torch.from_numpy(bins).to(device)
    return ResidentContig(index, length, d_cts, d_ref, d_core, d_db, d_gc, gc_start)


def walk_contig(caller, rc: ResidentContig, window: int = WINDOW, with_stats: bool = True, x: int = 1,
                all_positions: bool = False, stream=None, d_core=None) -> int:
    """Call contig `rc` window by window (asynchronous on `stream`); returns the number of windows."""
    n_block = rc.length
    core = rc.d_core if d_core is None else d_core
    k = 0
    if with_stats:  # the GC table follows the contig being walked (None switches it off)
        caller.set_gc_bins(None if rc.d_gc is None else rc.d_gc.data_ptr(), 0 if rc.d_gc is None else rc.d_gc.numel(), rc.gc_start)
    for first in range(0, n_block, window):
        n = min(window, n_block - first)
        lc, lr = min(2, first), min(4, first)
        caller.chain_device(rc.d_cts.data_ptr() + (first - lc) * PILEUP_BYTES, rc.d_ref.data_ptr() + (first - lr), x, n_block,
                            first, n, core.data_ptr() + first * CORE_BYTES, all_positions=all_positions,
                            d_dbsnp=None if rc.d_dbsnp is None else rc.d_dbsnp.data_ptr() + first, with_stats=with_stats,
                            stream=stream)
        k += 1
    return k


def walk_with_report(caller, contigs: Sequence[ResidentContig], names: Sequence[str], window: int, stream=None, **report_kw) -> str:
    """Walk the contigs in order and render the run's JSON report (bsc_report_json: the reference's output_stats): the
    totals, the per-contig copies of the seven [all, passed] pairs (differences of bsc_get_site_totals around each contig,
    the reference's gt_ctg_stats) and — for contigs made `with_gc` — the GC-by-coverage table."""
    from . import report

    per_contig = []
    before = caller.site_totals()
    for rc, name in zip(contigs, names):
        walk_contig(caller, rc, window, True, stream=stream)
        after = caller.site_totals()
        per_contig.append((name, after - before))
        before = after
    caller.set_gc_bins(None, 0, 0)
    gc = caller.gc_stats() if any(rc.d_gc is not None for rc in contigs) else None
    return report.render_json(caller.site_stats(), gc=gc, contigs=per_contig, **report_kw)
