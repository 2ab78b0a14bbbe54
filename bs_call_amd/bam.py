"""BAM in, blocks of templates out: `bsc_bam_*` (csrc/bamio.c, host C + zlib, no htslib) — the reference's reader thread
(read_input over get_next_align_details).  A block is what the reference queues for process_template_vector."""
import ctypes as C

import numpy as np

from . import _lib
from .abi import MISMS, RAW_TEMPLATE
from .caller import _check


class BamReader:
    def __init__(self, path, mapq_thresh=20, max_template_len=1000, keep_unmatched=False, ignore_duplicates=False, keep_duplicates=False,
                 threads=0, region=None):
        """threads: helper threads that inflate the BGZF blocks ahead of the parser (0 = the caller's thread does it).
        region: (tid, start, stop), 1-based inclusive: only the alignments overlapping it are read (the reference's -r)."""
        self._L = _lib.load()
        h = C.c_void_p()
        _check(self._L.bsc_bam_open_threads(str(path).encode(), int(threads), C.byref(h)))
        self._h = h
        reg = region or (0, 0, 0)
        self._par = _lib.ReaderParams(mapq_thresh, max_template_len, int(keep_unmatched), int(ignore_duplicates), int(keep_duplicates),
                                      int(reg[0]), int(reg[1]), int(reg[2]))

    def close(self):
        if self._h:
            self._L.bsc_bam_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def refs(self):
        """[(name, length)] of the @SQ list, in header order (a block's tid indexes it)."""
        n = self._L.bsc_bam_n_refs(self._h)
        return [(self._L.bsc_bam_ref_name(self._h, i).decode("utf-8", "replace"), int(self._L.bsc_bam_ref_len(self._h, i))) for i in range(n)]

    @property
    def header_text(self):
        return self._L.bsc_bam_header_text(self._h).decode("utf-8", "replace")

    def blocks(self):
        """Yields (tid, y, RAW_TEMPLATE[nr], read bytes, MISMS[]) per block, copies (the library's block lives until the
        next call)."""
        blk = _lib.ReadBlock()
        while True:
            r = self._L.bsc_bam_next_block(self._h, C.byref(self._par), C.byref(blk))
            _check(min(r, 0))
            if r == 0:
                return
            tpl = np.frombuffer(C.string_at(blk.tpl, blk.nr * RAW_TEMPLATE.itemsize), dtype=RAW_TEMPLATE).copy()
            seq = np.frombuffer(C.string_at(blk.seq, blk.seq_bytes), dtype=np.uint8).copy() if blk.seq_bytes else np.zeros(0, np.uint8)
            ms = np.frombuffer(C.string_at(blk.misms, blk.n_misms * MISMS.itemsize), dtype=MISMS).copy() if blk.n_misms else np.zeros(0, MISMS)
            yield int(blk.tid), int(blk.y), tpl, seq, ms

    def filter_counts(self):
        """(reads[15], bases[15]) by the reader's verdict (gt_filter_reason order; [14] = PairNotFound)."""
        cts, bases = (C.c_uint64 * 15)(), (C.c_uint64 * 15)()
        self._L.bsc_bam_filter_counts(self._h, cts, bases)
        return list(cts), list(bases)

    def malformed(self):
        """BAM records dropped because their CIGAR does not cover l_seq query bases (bsc_bam_malformed)."""
        return int(self._L.bsc_bam_malformed(self._h))


def fasta_contig(path, name, length_hint=0):
    """One contig of a FASTA file (plain / gzip / bgzip) as reference codes 0 = N, 1..4 = ACGT (bsc_fasta_contig)."""
    L = _lib.load()
    cap = int(length_hint) or (1 << 20)
    while True:
        buf = np.empty(cap, dtype=np.uint8)
        n = C.c_uint64(0)
        rc = L.bsc_fasta_contig(str(path).encode(), name.encode(), buf.ctypes.data, cap, C.byref(n))
        if rc == 0:
            return buf[: n.value].copy()
        if n.value > cap:
            cap = int(n.value)
            continue
        _check(rc)


def block_reference(codes, x, y):
    """work->ref1 of the block x .. y: the codes of x .. y + 2 as get_sequence_string hands them over (bsc_block_reference)."""
    L = _lib.load()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.empty(int(y) - int(x) + 3, dtype=np.uint8)
    _check(L.bsc_block_reference(codes.ctypes.data, codes.size, int(x), out.size, out.ctypes.data))
    return out
