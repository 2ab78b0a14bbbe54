"""File-fed sharding on the CPU (world_size-2 gloo): the ranks of a sharded run deal the header's contigs among themselves
(shard.assign_contigs), each reads the stretches of ONE BAM file that hold its contigs (bsc_bamstream_open_contigs: a binary search over the
file's BGZF blocks, no index file) and forms their blocks (the device reader's statements, run by tests/emul on the CPU); the ranks' blocks
are the single reader's blocks contig by contig, their filter counters add up to its counters through the all-reduce the pipeline uses.
(The GPU rehearsal — the real kernels, BCF and report bytes — is tests/test_gpu_shard_bam.py.)"""
import importlib.util
import os
import pickle
import socket
import sys

import numpy as np
import pytest

from bs_call_amd import shard
from bs_call_amd.bamdev import BamStream
from bs_call_amd.caller import BscError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


REFS = [("chr1", 60_000), ("chr2", 90_000), ("chr3", 30_000), ("chr4", 45_000), ("chrEmpty", 20_000)]


def make_file(path, seed=5, aligned=True, block=4000):
    TE = _load("emul_mod_s", "tests/test_bamdev_emul.py")
    rng = np.random.default_rng(seed)
    recs = []
    for tid in (0, 1, 2, 3):
        part = TE._sane_records(rng, 700 if tid != 2 else 40, dup_rate=0.1)  # chr3's few records fit inside one block
        for r in part:
            r["tid"] = r["mtid"] = tid
            r["name"] = "c%d_%s" % (tid, r["name"])
        recs += part
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    recs += [TE.TB.rec("unplaced%d" % i, 4, -1, -1, tid=-1, mtid=-1) for i in range(3)]  # unmapped reads without a position: the file's end
    TE.W.write_bam(path, REFS, recs, block=block, aligned=aligned)
    return TE


def stream_of_contigs(path, contigs, threads=3):
    data, offs = bytearray(), []
    with BamStream(path, threads=threads, slab_bytes=65536, n_slabs=3, contigs=contigs) as s:
        n_ref = len(s.refs)
        for off, b, ro, last in s.slabs():
            assert off == len(data)
            data += b
            offs += [off + int(v) for v in ro]
    return bytes(data), np.array(offs, dtype=np.uint64), n_ref


def _worker(rank, world, port, path, out_path):
    sys.path.insert(0, ROOT)
    import ctypes
    import subprocess
    import tempfile

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    TE = _load("emul_mod_w%d" % rank, "tests/test_bamdev_emul.py")
    so = os.path.join(tempfile.mkdtemp(), "libemul.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "emul", "bamdev_emul.cpp")])
    lib = ctypes.CDLL(so)
    lib.bd_emul_error.restype = ctypes.c_char_p
    lib.bd_emul_n_blocks.restype = ctypes.c_uint64
    parts = shard.assign_contigs([l for _, l in REFS], world)
    mine = sorted(parts[rank]) + ([-1] if len(REFS) - 1 in parts[rank] else [])
    stream = stream_of_contigs(path, mine)
    (blocks, cts, bases), info = TE.emul_blocks(lib, path, stream=stream, contigs=mine)
    total = shard.allreduce_counts(np.array(cts + bases, dtype=np.uint64))
    with open("%s.%d" % (out_path, rank), "wb") as f:
        pickle.dump({"mine": mine, "blocks": blocks, "cts": cts, "total": total.tolist(), "stream_bytes": len(stream[0]), "records": len(stream[1])}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_one_file(tmp_path):
    import torch.multiprocessing as mp

    path = str(tmp_path / "multi.bam")
    TE = make_file(path)
    want, cts, bases = TE.TB.c_blocks(path)  # the single reader (csrc/bamio.c)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(2, port, path, out), nprocs=2, join=True)
    res = [pickle.load(open("%s.%d" % (out, r), "rb")) for r in range(2)]
    assert sorted(t for r in res for t in r["mine"]) == [-1, 0, 1, 2, 3, 4]
    got = sorted((b for r in res for b in r["blocks"]), key=lambda b: (b[0], b[1]))
    assert got == sorted(want, key=lambda b: (b[0], b[1]))  # contig by contig the single reader's blocks
    for r in res:
        assert {b[0] for b in r["blocks"]} <= set(r["mine"])
        assert r["total"] == cts + bases  # the all-reduced counters are the single reader's
    # no rank read the whole file
    whole = stream_of_contigs(path, None)
    assert all(r["stream_bytes"] < len(whole[0]) for r in res)


def test_selections_stretches_and_refusals(tmp_path):
    path = str(tmp_path / "multi.bam")
    TE = make_file(path, seed=9)
    import ctypes
    import subprocess

    so = str(tmp_path / "libemul.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "emul", "bamdev_emul.cpp")])
    lib = ctypes.CDLL(so)
    lib.bd_emul_error.restype = ctypes.c_char_p
    lib.bd_emul_n_blocks.restype = ctypes.c_uint64
    want, cts, bases = TE.TB.c_blocks(path)
    by_tid = lambda blocks, t: [b for b in blocks if b[0] == t]
    sums = np.zeros(30, dtype=np.uint64)
    for sel in ([0], [1], [2], [3], [4, -1]):  # every contig alone (chr3 lies inside one block; chrEmpty has no records), the unplaced reads with the last
        (blocks, c1, b1), _ = TE.emul_blocks(lib, path, stream=stream_of_contigs(path, sel), contigs=sel)
        assert blocks == [b for t in sel for b in by_tid(want, t)]
        sums += np.array(c1 + b1, dtype=np.uint64)
    assert sums.tolist() == cts + bases
    for sel in ([0, 2], [3, 1], [0, 1, 2, 3, 4, -1], []):  # stretches apart, touching, everything, nothing
        (blocks, _, _), _ = TE.emul_blocks(lib, path, stream=stream_of_contigs(path, sel), contigs=sel)
        assert blocks == [b for t in sorted(set(sel) - {-1}) for b in by_tid(want, t)]
    # records cut by block boundaries (htsjdk's writer; here blocks of 777, 100 and 5000 bytes — a record over several blocks, several records in
    # one): the first record START of a block is found (a chain of plausible record headers), a stretch begins there and ends with the bytes of
    # the following block(s) up to their first record start — the same blocks and counters as from the file whose blocks start at records
    for block in (777, 100, 5000):
        cut = str(tmp_path / ("cut%d.bam" % block))
        make_file(cut, seed=9, aligned=False, block=block)
        want_c, cts_c, bases_c = TE.TB.c_blocks(cut)
        assert want_c == want and cts_c == cts and bases_c == bases
        sums = np.zeros(30, dtype=np.uint64)
        whole_bytes = len(stream_of_contigs(cut, None)[0])
        for sel in ([0], [1], [2], [3], [4, -1]):
            st = stream_of_contigs(cut, sel)
            (blocks, c1, b1), _ = TE.emul_blocks(lib, cut, stream=st, contigs=sel)
            assert blocks == [b for t in sel for b in by_tid(want, t)], (block, sel)
            sums += np.array(c1 + b1, dtype=np.uint64)
            assert len(st[0]) < whole_bytes
        assert sums.tolist() == cts + bases, block
        for sel in ([0, 2], [3, 1], [0, 1, 2, 3, 4, -1], []):
            (blocks, _, _), _ = TE.emul_blocks(lib, cut, stream=stream_of_contigs(cut, sel), contigs=sel)
            assert blocks == [b for t in sorted(set(sel) - {-1}) for b in by_tid(want, t)], (block, sel)
        assert len(stream_of_contigs(cut, None)[1]) == len(stream_of_contigs(path, None)[1])
