"""The N > 1 path on CPU: partitioning (no site lost or duplicated) and the stats all-reduce over a
world_size-2 gloo group.  The per-rank counter blocks are produced by the CPU oracle here (checker role only);
on GPUs they come from SiteCaller.stats_vector()."""
import os
import socket
import sys

import numpy as np
import pytest

from bs_call_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_assignment_human_contigs():
    for ws in (1, 2, 4, 8):
        parts = shard.assign_contigs(shard.HUMAN_CONTIGS, ws)
        assert sorted(i for p in parts for i in p) == list(range(24))
        loads = [sum(shard.HUMAN_CONTIGS[i] for i in p) for p in parts]
        assert max(loads) / (sum(loads) / ws) < 1.08  # LPT balance on the human genome
        shard.check_partition(shard.HUMAN_CONTIGS, ws)
        shard.check_partition(shard.HUMAN_CONTIGS, ws, split_contigs=True)


def test_ragged_and_empty_partitions():
    shard.check_partition([1, 5, 4 << 20, (4 << 20) + 1, 0, 17], 3, window=1 << 20)
    shard.check_partition([10], 4)  # more ranks than contigs: some ranks idle
    assert shard.rank_windows([10], 3, 4) == []
    assert shard.windows_of(0, 0) == []
    with pytest.raises(ValueError):
        shard.assign_contigs([1], 0)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import bs_call_amd as B
    from bs_call_amd.abi import GT_HET
    from oracle import loader as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = [30_000, 50_000, 20_000, 40_000, 10_000]
    tb = O.Tables()
    per_contig = {}
    total = np.zeros(shard.STATS_WORDS, dtype=np.int64)
    for w in shard.rank_windows(lengths, rank, world, window=16_384):
        first = sum(lengths[: w.contig]) + w.start  # contigs laid end to end on the synthetic genome
        pile, ref = B.synth_pileup_host(777, first, w.length, 10)
        out, skip = O.call_sites(pile, ref, tb, O.LIBM, 1)
        cov = skip == 0
        v = np.array([w.length, int(cov.sum())] + np.bincount(out["max_gt"][cov], minlength=10).tolist()
                     + [int(GT_HET[out["max_gt"]][cov].sum())], dtype=np.int64)
        per_contig[w.contig] = per_contig.get(w.contig, 0) + v
        total += v
    allsum = shard.allreduce_stats(total)
    table = shard.gather_contig_stats(per_contig, len(lengths))
    # the printer's statistics (bsc_site_stats), one contig at a time on the rank that owns it
    site = np.zeros(1, dtype=B.SITE_STATS)
    for ci in shard.assign_contigs(lengths, world)[rank]:
        first = sum(lengths[:ci])
        pile, ref = B.synth_pileup_host(777, first, lengths[ci] + 2, 10)
        out, skip = O.call_sites(pile[: lengths[ci]], ref[: lengths[ci]], tb, O.LIBM, 1)
        O.vcf_block_stats(out, skip, ref, 1, site, np.zeros(2, dtype=np.uint32), tb.lfact_store)
    site_sum = shard.allreduce_site_stats(site[0])
    gc = np.random.default_rng(100 + rank).integers(0, 2**40, (4096, 101)).astype(np.uint64)  # a rank's GC-by-coverage table
    gc_sum = shard.allreduce_counts(gc)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, total, allsum, table, site[0].tobytes(), site_sum.tobytes(), int(gc_sum.sum()), gc_sum[17, 50]))


def test_world2_gloo_stats_allreduce():
    import torch.multiprocessing as mp

    import bs_call_amd as B

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, t0, a0, tab0, s0, ss0, g0, c0), (_, t1, a1, tab1, s1, ss1, g1, c1) = res
    gcs = [np.random.default_rng(100 + r).integers(0, 2**40, (4096, 101)).astype(np.uint64) for r in range(2)]
    assert g0 == g1 == int(gcs[0].sum() + gcs[1].sum()) and c0 == c1 == gcs[0][17, 50] + gcs[1][17, 50]
    s0, s1, ss0, ss1 = (np.frombuffer(b, dtype=B.SITE_STATS)[0] for b in (s0, s1, ss0, ss1))
    assert ss0.tobytes() == ss1.tobytes()
    for f in B.SITE_STATS.names:
        if f.endswith("_meth"):
            assert np.allclose(ss0[f], s0[f] + s1[f], rtol=1e-15)
        else:
            assert (ss0[f] == s0[f] + s1[f]).all(), f
    assert ss0["snps"][0] > 1000 and ss0["cov"][:, 0].sum() > 100_000
    assert (a0 == a1).all() and (a0 == t0 + t1).all() and (tab0 == tab1).all()
    assert a0[0] == 150_000 and (tab0.sum(axis=0) == a0).all()
    # same numbers as a single-rank run over the whole genome
    sys.path.insert(0, ROOT)
    import bs_call_amd as B
    from oracle import loader as O

    pile, ref = B.synth_pileup_host(777, 0, 150_000, 10)
    out, skip = O.call_sites(pile, ref, O.Tables(), O.LIBM, -4)
    cov = skip == 0
    assert a0[1] == cov.sum() and a0[2:12].tolist() == np.bincount(out["max_gt"][cov], minlength=10).tolist()
