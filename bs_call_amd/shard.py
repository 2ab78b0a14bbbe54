"""Multi-GPU sharding of the calling path.

Sites are independent inside the path and contigs are independent end to end (the reference itself is run
one process per contig and its outputs concatenated: reference README.md:73-76, src/process_sam_header.c:67-70),
so the work is PARTITIONED: every rank (one process per GPU) calls its own contigs / position windows and no
rank needs another rank's data.  There is no data-path collective.  The only exchange is at the end of a run:
the per-rank counter blocks (SiteCaller.stats_vector(), the sums the reference keeps in bs_stats /
gt_ctg_stats, include/bs_call.h:75-85,124-146) are summed with one all-reduce — RCCL when the ranks own GPUs,
gloo in the CPU tests.

Host logic only; nothing here computes calls.
"""
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

WINDOW = 4 << 20  # positions per window (fixed 4 Mi-site chunks)
STATS_WORDS = 13  # sites, covered, gt_hist[10], het_calls


@dataclass(frozen=True)
class Window:
    contig: int  # index into the contig list
    start: int  # first position (0-based offset inside the contig)
    length: int


def assign_contigs(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time assignment of whole contigs to ranks: contigs sorted by length (ties by index),
    each given to the currently least loaded rank (ties to the lowest rank).  Deterministic on every rank."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(lengths[i])
    return out


def windows_of(contig: int, length: int, window: int = WINDOW) -> List[Window]:
    """Split one contig into fixed windows (the last one ragged).  Calls are per-site, so windows need no halo
    for this path (the 2-site halo of SURVEY 8e belongs to the VCF printer's context window)."""
    return [Window(contig, s, min(window, length - s)) for s in range(0, length, window)]


def rank_windows(lengths: Sequence[int], rank: int, world_size: int, window: int = WINDOW,
                 split_contigs: bool = False) -> List[Window]:
    """The windows rank `rank` calls.  Default: whole contigs by LPT.  split_contigs=True deals the windows of
    all contigs round-robin in (contig, start) order instead — finer balance when contigs are few."""
    if split_contigs:
        allw = [w for c, n in enumerate(lengths) for w in windows_of(c, int(n), window)]
        return allw[rank::world_size]
    mine = assign_contigs(lengths, world_size)[rank]
    return [w for c in sorted(mine) for w in windows_of(c, int(lengths[c]), window)]


def check_partition(lengths: Sequence[int], world_size: int, window: int = WINDOW, split_contigs: bool = False) -> None:
    """Every position of every contig belongs to exactly one rank's windows."""
    seen: Dict[Tuple[int, int], int] = {}
    for r in range(world_size):
        for w in rank_windows(lengths, r, world_size, window, split_contigs):
            key = (w.contig, w.start)
            if key in seen:
                raise AssertionError("window %r assigned to ranks %d and %d" % (key, seen[key], r))
            seen[key] = r
    for c, n in enumerate(lengths):
        covered = sum(min(window, int(n) - s) for (cc, s) in seen if cc == c)
        if covered != int(n):
            raise AssertionError("contig %d: %d of %d positions assigned" % (c, covered, n))


def allreduce_stats(local_stats: np.ndarray, device=None) -> np.ndarray:
    """Sum the per-rank counter blocks over the default process group (no-op without one).  `device`: a torch
    device for the reduction buffer ('cuda:N' under RCCL; None/'cpu' under gloo)."""
    import torch
    import torch.distributed as dist

    v = np.asarray(local_stats, dtype=np.int64)
    if v.shape != (STATS_WORDS,):
        raise ValueError("stats vector must have %d words" % STATS_WORDS)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v.copy()
    t = torch.from_numpy(v.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allreduce_site_stats(local, device=None):
    """Sum the per-rank bsc_site_stats blocks (a SITE_STATS record, SiteCaller.site_stats()) over the default process
    group: every field is a sum over positions, so the shards of a run add.  Two collectives —
    the u64 counters as one int64 vector, the methylation profiles as one float64 vector (< 250 KB together:
    latency-bound, the link bandwidth does not matter).  No-op without a process group."""
    import torch
    import torch.distributed as dist

    from .abi import SITE_STATS, SITE_STATS_INT_WORDS

    rec = np.array(local, dtype=SITE_STATS).reshape(1).copy()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec[0]
    raw = rec.view(np.uint8).reshape(-1)
    ints = raw[: SITE_STATS_INT_WORDS * 8].view(np.int64)
    flts = raw[SITE_STATS_INT_WORDS * 8 :].view(np.float64)
    ti, tf = torch.from_numpy(ints.copy()), torch.from_numpy(flts.copy())
    if device is not None:
        ti, tf = ti.to(device), tf.to(device)
    dist.all_reduce(ti, op=dist.ReduceOp.SUM)
    dist.all_reduce(tf, op=dist.ReduceOp.SUM)
    ints[:] = ti.cpu().numpy()
    flts[:] = tf.cpu().numpy()
    return rec[0]


def allreduce_counts(local: np.ndarray, device=None) -> np.ndarray:
    """Sum a table of u64 counts over the default process group: the GC-by-coverage table (SiteCaller.gc_stats()), the
    non-CpG read profile (ReadProfile.counts, padded to a common length by the caller), the reader's filter counters —
    everything else the report needs that is a sum.  No-op without a process group."""
    import torch
    import torch.distributed as dist

    v = np.ascontiguousarray(local, dtype=np.uint64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v.copy()
    t = torch.from_numpy(v.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().view(np.uint64).reshape(v.shape)


def gather_contig_stats(per_contig: Dict[int, np.ndarray], n_contigs: int, device=None) -> np.ndarray:
    """Per-contig counter blocks -> the full [n_contigs, STATS_WORDS] table on every rank.  Each contig is owned
    by one rank, so a sum-all-reduce of the zero-padded table is a gather."""
    import torch
    import torch.distributed as dist

    tab = np.zeros((n_contigs, STATS_WORDS), dtype=np.int64)
    for c, v in per_contig.items():
        tab[c] += np.asarray(v, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tab
    t = torch.from_numpy(tab)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


# human-scale contig lengths (GRCh38 primary chromosomes 1-22, X, Y) for config 3 of BASELINE.json
HUMAN_CONTIGS = [
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
    135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
    46709983, 50818468, 156040895, 57227415,
]
