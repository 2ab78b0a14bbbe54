"""The reader on the device: `bsc_bamstream_*` (csrc/bamstream.c: the BAM file as inflated bytes in page-locked slabs + record
offsets) and `bsc_bamdev_*` (csrc/bamdev.hip: the blocks of templates formed in HBM).  ctypes mirrors for the tests, bench.py and
pipeline.run(device_reader=True)."""
import ctypes as C

import numpy as np

from . import _lib
from .abi import MISMS, RAW_TEMPLATE
from .caller import _check


class BamStream:
    """The host half alone (no GPU needed): slabs of inflated bytes with the records' offsets."""

    def __init__(self, path, threads=0, slab_bytes=0, n_slabs=0, contigs=None):
        """contigs: None = the whole file; a list of tids (-1 = the unplaced reads at the end): the stretches of the file that hold them"""
        self._L = _lib.load()
        h = C.c_void_p()
        if contigs is None:
            _check(self._L.bsc_bamstream_open(str(path).encode(), int(threads), int(slab_bytes), int(n_slabs), C.byref(h)))
        else:
            t = np.ascontiguousarray(contigs, dtype=np.int32)
            _check(self._L.bsc_bamstream_open_contigs(str(path).encode(), int(threads), int(slab_bytes), int(n_slabs), t.ctypes.data, len(t), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self._L.bsc_bamstream_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def refs(self):
        n = self._L.bsc_bamstream_n_refs(self._h)
        return [(self._L.bsc_bamstream_ref_name(self._h, i).decode("utf-8", "replace"), int(self._L.bsc_bamstream_ref_len(self._h, i))) for i in range(n)]

    @property
    def header_text(self):
        return self._L.bsc_bamstream_header_text(self._h).decode("utf-8", "replace")

    @property
    def first_record(self):
        return int(self._L.bsc_bamstream_first_record(self._h))

    @property
    def threads(self):
        return int(self._L.bsc_bamstream_threads(self._h))

    def slabs(self):
        """Yields (stream_off, bytes (copy), record offsets relative to the slab (copy), last)."""
        sl = _lib.BamSlab()
        while True:
            r = self._L.bsc_bamstream_next(self._h, C.byref(sl))
            _check(min(r, 0))
            if r == 0:
                return
            data = C.string_at(sl.bytes, sl.n_bytes)
            offs = np.frombuffer(C.string_at(sl.rec_off, sl.n_recs * 4), dtype=np.uint32).copy() if sl.n_recs else np.zeros(0, np.uint32)
            out = (int(sl.stream_off), data, offs, bool(sl.last))
            _check(self._L.bsc_bamstream_release(self._h, C.byref(sl)))
            yield out


def drain_stream(path, threads=0, slab_bytes=0, n_slabs=0):
    """The streamer alone, nothing copied: (inflated bytes, records, seconds, helper threads) — the host's inflate rate."""
    import time

    L = _lib.load()
    h = C.c_void_p()
    t0 = time.time()
    _check(L.bsc_bamstream_open(str(path).encode(), int(threads), int(slab_bytes), int(n_slabs), C.byref(h)))
    sl = _lib.BamSlab()
    nb = nr = 0
    try:
        nth = int(L.bsc_bamstream_threads(h))
        while True:
            r = L.bsc_bamstream_next(h, C.byref(sl))
            _check(min(r, 0))
            if r == 0:
                break
            nb += sl.n_bytes
            nr += sl.n_recs
            _check(L.bsc_bamstream_release(h, C.byref(sl)))
    finally:
        L.bsc_bamstream_close(h)
    return nb, nr, time.time() - t0, nth


class DeviceBamReader:
    """bsc_bamdev_*: blocks of raw templates formed on the device of `caller` (a SiteCaller)."""

    def __init__(self, caller, path, mapq_thresh=20, max_template_len=1000, keep_unmatched=False, ignore_duplicates=False, keep_duplicates=False,
                 threads=0, region=None, contigs=None):
        """contigs: None = the whole file; a list of tids (-1 = the unplaced reads): one rank's share of a sharded run"""
        self._L = _lib.load()
        self._c = caller
        h = C.c_void_p()
        if contigs is None:
            _check(self._L.bsc_bamdev_open(caller._h, str(path).encode(), int(threads), C.byref(h)))
        else:
            t = np.ascontiguousarray(contigs, dtype=np.int32)
            _check(self._L.bsc_bamdev_open_contigs(caller._h, str(path).encode(), int(threads), t.ctypes.data, len(t), C.byref(h)))
        self._h = h
        reg = region or (0, 0, 0)
        self._par = _lib.ReaderParams(mapq_thresh, max_template_len, int(keep_unmatched), int(ignore_duplicates), int(keep_duplicates),
                                      int(reg[0]), int(reg[1]), int(reg[2]))

    def close(self):
        if self._h:
            self._L.bsc_bamdev_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def refs(self):
        n = self._L.bsc_bamdev_n_refs(self._h)
        return [(self._L.bsc_bamdev_ref_name(self._h, i).decode("utf-8", "replace"), int(self._L.bsc_bamdev_ref_len(self._h, i))) for i in range(n)]

    @property
    def header_text(self):
        return self._L.bsc_bamdev_header_text(self._h).decode("utf-8", "replace")

    def device_blocks(self):
        """Yields the DevReadBlock of every block: device pointers, valid until the next one is asked for."""
        blk = _lib.DevReadBlock()
        while True:
            r = self._L.bsc_bamdev_next_block(self._h, C.byref(self._par), C.byref(blk))
            _check(min(r, 0))
            if r == 0:
                return
            yield blk

    def fetch(self, blk):
        tpl = np.zeros(blk.nr, dtype=RAW_TEMPLATE)
        seq = np.zeros(max(1, blk.seq_bytes), dtype=np.uint8)
        ms = np.zeros(max(1, blk.n_misms), dtype=MISMS)
        _check(self._L.bsc_bamdev_fetch_block(self._h, C.byref(blk), tpl.ctypes.data, seq.ctypes.data, ms.ctypes.data))
        return tpl, seq[: blk.seq_bytes], ms[: blk.n_misms]

    def blocks(self):
        """BamReader.blocks()'s view: (tid, y, RAW_TEMPLATE[nr], read bytes, MISMS[]) per block, on the host."""
        for blk in self.device_blocks():
            tpl, seq, ms = self.fetch(blk)
            yield int(blk.tid), int(blk.y), tpl, seq, ms

    def filter_counts(self):
        cts, bases = (C.c_uint64 * 15)(), (C.c_uint64 * 15)()
        _check(self._L.bsc_bamdev_filter_counts(self._h, cts, bases))
        return list(cts), list(bases)

    def malformed(self):
        return int(self._L.bsc_bamdev_malformed(self._h))

    def run_stats(self):
        c, s = (C.c_uint64 * 4)(), (C.c_double * 2)()
        self._L.bsc_bamdev_run_stats(self._h, c, s)
        return {"passes": int(c[0]), "replay_passes": int(c[1]), "records": int(c[2]), "bytes_uploaded": int(c[3]), "wait_s": float(s[0]), "device_s": float(s[1])}
