"""Synthetic L-reads blocks at config sizes: the host generator (csrc/synth_reads.c) run over chunks of the
block on several threads, concatenated into one template list + one read buffer.  Bench / test support only."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from .abi import TEMPLATE


def synth_block(seed, x, n_sites, coverage, flags=0, chunk=1_000_000, threads=None):
    """Templates (in the order of their start positions) and read bytes of one block over positions x .. x + n_sites - 1
    -> (TEMPLATE[nt], uint8 seq, y) with y = the last position a read covers (the block's right end, as
    process_template_vector finds it: src/process_template.c:24-28).  The synthetic genome is the one of the L-pileup
    generator (site index = position); every chunk is generated independently (a pair that starts in one chunk and whose
    mate starts in the next loses that mate, like a pair cut by a block boundary)."""
    L = _lib.load()
    threads = threads or min(os.cpu_count() or 1, 32)
    starts = list(range(0, n_sites, chunk))

    def one(s0):
        m = min(chunk, n_sites - s0)
        max_t = m * coverage // 150 + 64
        cap = max_t * 200 + 1024
        tpl = np.zeros(max_t, dtype=TEMPLATE)
        seq = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        nt = L.bsc_synth_reads_host(seed, x + s0, m, coverage, flags, tpl.ctypes.data_as(C.c_void_p), max_t,
                                    seq.ctypes.data_as(C.c_void_p), cap, C.byref(used))
        if nt < 0:
            raise RuntimeError("bsc_synth_reads_host: buffer too small")
        return tpl[:nt], seq[: used.value]

    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(one, starts))
    nt = sum(len(p[0]) for p in parts)
    nb = sum(len(p[1]) for p in parts)
    tpl = np.zeros(nt, dtype=TEMPLATE)
    seq = np.zeros(nb, dtype=np.uint8)
    t0 = b0 = 0
    for pt, ps in parts:
        tpl[t0 : t0 + len(pt)] = pt
        tpl["off"][t0 : t0 + len(pt)] += np.uint64(b0)
        seq[b0 : b0 + len(ps)] = ps
        t0 += len(pt)
        b0 += len(ps)
    y = int((tpl["pos"].astype(np.int64) + tpl["len"]).max()) - 1
    return tpl, seq, y


def algorithmic_bytes_in(tpl, seq):
    """1 byte per base + 16 bytes per template."""
    return int(seq.size) + 16 * int(len(tpl))


TPL_WALK_KNOWN, TPL_WALKED0 = 1, 2  # bsc_template.flags (include/bscall_amd.h: BSC_TPL_*)


def walk_flags(tpl, seq):
    """bsc_template_walk_flags() of every template (include/bscall_amd.h): BSC_TPL_WALK_KNOWN, and BSC_TPL_WALKED0 where read 0
    holds a base whose quality is neither 0 nor 63 (what decides whether HOT LOOP A flips the orientation for read 1,
    src/call_genotypes.c:198-211,224).  Plain numpy over the read bytes: for tests and for callers that build templates in
    Python."""
    q = np.asarray(seq, dtype=np.uint8) >> 2
    ok = np.concatenate(([0], np.cumsum((q != 0) & (q != 63), dtype=np.int64)))
    a = tpl["off"][:, 0].astype(np.int64)
    b = a + tpl["len"][:, 0].astype(np.int64)
    walked = (tpl["len"][:, 0] > 0) & (ok[np.minimum(b, len(q))] - ok[np.minimum(a, len(q))] > 0)
    return (TPL_WALK_KNOWN | np.where(walked, TPL_WALKED0, 0)).astype(np.uint32)
