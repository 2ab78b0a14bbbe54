"""Host-side mirror of bs_call's calc API over the C ABI (include/bscall_amd.h).

`SiteCaller` plays the role of the reference's calc-thread pool: create it once (init_calc_threads +
fill_base_prob_table, src/process.c:165-166), feed it blocks of pile-ups (the call_thread loop,
src/call_genotypes.c:43-115), close it at the end (join_calc_threads).  Everything is computed by the
gfx950 kernels; nothing here computes on the CPU or uses the CPU checker.
"""
import ctypes as C

import numpy as np

from . import _lib
from .abi import GT_METH, PILEUP, SITE_STATS, TEMPLATE, VCF_CORE, VCF_REC

SYNTH_NRUNS = 1


class BscError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bscall_amd error %d: %s" % (code, msg))
        self.code = code


class BscInexactWarning(UserWarning):
    """Some pile-up sums left the range where the reference's float sums are exact (BSC_WARN_INEXACT)."""


def _check(rc):
    if rc == 1:  # BSC_WARN_INEXACT: results are written
        import warnings

        warnings.warn(_lib.load().bsc_last_error().decode("utf-8", "replace"), BscInexactWarning, stacklevel=3)
        return
    if rc != 0:
        raise BscError(rc, _lib.load().bsc_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _check_dest(name, a, nbytes):
    """A caller-supplied destination the library writes `nbytes` bytes into: the C ABI takes a bare pointer, so an
    undersized or strided array would be a host heap overflow."""
    if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"] or not a.flags["WRITEABLE"]:
        raise ValueError("%s must be a writable C-contiguous numpy array" % name)
    if a.nbytes != nbytes:
        raise ValueError("%s has %d bytes, the call writes %d" % (name, a.nbytes, nbytes))


class SiteCaller:
    def __init__(self, under_conv=0.01, over_conv=0.05, ref_bias=2.0, min_qual=20, device=-1):
        self._L = _lib.load()
        p = _lib.Params(under_conv, over_conv, ref_bias, min_qual, device)
        self.params = {"under_conv": float(under_conv), "over_conv": float(over_conv), "ref_bias": float(ref_bias), "min_qual": int(min_qual)}
        h = C.c_void_p()
        _check(self._L.bsc_create(C.byref(p), C.byref(h)))
        self._h = h
        self._pending = None  # the submitted, not yet fetched block (block_submit / block_submit_to)

    # -- lifecycle ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.bsc_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- tables -----------------------------------------------------------------------------------
    def tables(self):
        q = np.zeros((44, 5), dtype=np.float64)
        lf = np.zeros(256, dtype=np.float64)
        _check(self._L.bsc_get_tables(self._h, _ptr(q), _ptr(lf)))
        return q, lf

    # -- host blocks ------------------------------------------------------------------------------
    def call_sites(self, pile, ref, out_stride=200, out=None, skip=None):
        """pile: PILEUP[n], ref: uint8[n] codes 0..4 -> (GT_METH[n] or raw uint8[n, stride], skip uint8[n]).
        `out` / `skip` may be preallocated (e.g. pinned, see pinned_empty) arrays of the right size."""
        pile = np.ascontiguousarray(pile, dtype=PILEUP)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = len(pile)
        if len(ref) != n:
            raise ValueError("pile and ref differ in length")
        if out is None:
            out = np.zeros(n, dtype=GT_METH) if out_stride == 200 else np.zeros((n, out_stride), dtype=np.uint8)
        if skip is None:
            skip = np.zeros(n, dtype=np.uint8)
        _check_dest("out", out, n * out_stride)
        _check_dest("skip", skip, n)
        _check(self._L.bsc_call_sites(self._h, _ptr(pile), _ptr(ref), n, _ptr(out), out_stride, _ptr(skip)))
        return out, skip

    # -- reads -> pile-up (HOT LOOP A) and whole blocks ------------------------------------------------
    def accumulate(self, templates, seq, x, y, out=None):
        """templates: TEMPLATE[nr]; seq: uint8 read bytes; positions x..y inclusive -> PILEUP[y-x+1].
        `out`: optional preallocated PILEUP array (reusing one avoids first-touch page faults in timing loops)."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        n = max(int(y) - int(x) + 1, 0)
        if out is None:
            out = np.zeros(max(n, 1), dtype=PILEUP)
        elif out.dtype != PILEUP or not out.flags["C_CONTIGUOUS"] or len(out) < n:
            raise ValueError("out must be a C-contiguous PILEUP array of at least y - x + 1 = %d records" % n)
        _check(self._L.bsc_accumulate(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(out)))
        return out[: max(int(y) - int(x) + 1, 0)]

    def accumulate_device(self, d_tpl, nr, d_seq, seq_bytes, x, y, d_cts, stream=None):
        """HOT LOOP A on device-resident reads (raw device pointers); asynchronous — block_status() collects the verdict."""
        _check(self._L.bsc_accumulate_device(self._h, d_tpl, nr, d_seq, seq_bytes, x, y, d_cts, stream))

    def block_status(self, stream=None):
        """Wait for `stream` and raise / warn as the block queued last deserves (invalid template, inexact sums)."""
        _check(self._L.bsc_block_status(self._h, stream))

    def last_accumulate_ms(self):
        ms = C.c_float()
        _check(self._L.bsc_last_accumulate_ms(self._h, C.byref(ms)))
        return ms.value

    def call_block(self, templates, seq, x, y, ref, out_stride=200, out=None, skip=None):
        """One call_genotypes_ML block: accumulate + call.  ref: uint8[y-x+1] codes -> (GT_METH[..], skip)."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n:
            raise ValueError("ref must have y - x + 1 entries")
        if out is None:
            out = np.zeros(n, dtype=GT_METH) if out_stride == 200 else np.zeros((n, out_stride), dtype=np.uint8)
        if skip is None:
            skip = np.zeros(n, dtype=np.uint8)
        _check_dest("out", out, n * out_stride)
        _check_dest("skip", skip, n)
        _check(self._L.bsc_call_block(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(ref),
                                      _ptr(out), out_stride, _ptr(skip)))
        return out, skip

    def block_submit(self, templates, seq, x, y, ref, out_stride=200):
        """Queue one block (accumulate + call) and return at once; the inputs may be reused immediately."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        if len(ref) != int(y) - int(x) + 1:
            raise ValueError("ref must have y - x + 1 entries")
        _check(self._L.bsc_block_submit(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(ref), out_stride))
        self._pending = (int(y) - int(x) + 1, out_stride)

    def block_submit_to(self, templates, seq, x, y, ref, out, skip):
        """block_submit with the destination named up front (pinned arrays, see PinnedBuffer): the copy-out is queued
        behind the kernels; block_fetch() then only waits, reports and returns (out, skip)."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n:
            raise ValueError("ref must have y - x + 1 entries")
        stride = out.dtype.itemsize if out.ndim == 1 else out.shape[1]
        _check_dest("out", out, n * stride)
        _check_dest("skip", skip, n)
        _check(self._L.bsc_block_submit_to(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(ref),
                                           _ptr(out), stride, _ptr(skip)))
        self._pending = (out, skip)

    def blocks_submit_to(self, blocks, ref, out_stride=200, inplace=False):
        """bsc_blocks_submit_to: blocks = [(templates, seq, x, y), ...], ref = their y - x + 3 codes each (list of arrays); the
        images of all blocks land in ONE page-locked array (block b from position offset off[b] on, a multiple of 64).  Returns
        (off, out, skip) after the fetch: out = GT_METH[P] (stride 200) or uint8[P, 208].  inplace: bsc_blocks_submit_to_inplace
        from page-locked copies of the joined inputs (no staging copy inside the library)."""
        from .abi import BLOCK_DESC

        tpls, seqs, desc, o = [], [], np.zeros(len(blocks), dtype=BLOCK_DESC), 0
        for i, (t, sq, x, y) in enumerate(blocks):
            t = np.array(t, dtype=TEMPLATE)
            sq = np.ascontiguousarray(sq, dtype=np.uint8)
            t["off"] += np.uint64(o)
            o += sq.size
            tpls.append(t)
            seqs.append(sq)
            desc[i] = (x, y, len(t), 0)
        tpl, seq = np.concatenate(tpls), np.concatenate(seqs)
        refs = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint8) for r in ref]))
        P = sum(((int(y) - int(x) + 1 + 63) // 64) * 64 for _, _, x, y in blocks)
        self._pin = (PinnedBuffer(P, GT_METH) if out_stride == 200 else PinnedBuffer((P, out_stride), np.uint8), PinnedBuffer(P, np.uint8))
        out, skip = self._pin[0].array, self._pin[1].array
        off = np.zeros(len(blocks), dtype=np.uint64)
        fn = self._L.bsc_blocks_submit_to
        if inplace:
            fn = self._L.bsc_blocks_submit_to_inplace
            pins = (PinnedBuffer(len(tpl), TEMPLATE), PinnedBuffer(max(seq.size, 1), np.uint8), PinnedBuffer(max(refs.size, 1), np.uint8))
            pins[0].array[:] = tpl
            pins[1].array[: seq.size] = seq
            pins[2].array[: refs.size] = refs
            tpl, seq_a, refs = pins[0].array, pins[1].array, pins[2].array
            _check(fn(self._h, _ptr(desc), len(desc), _ptr(tpl), _ptr(seq_a), seq.size, _ptr(refs), _ptr(out), out_stride, _ptr(skip), _ptr(off)))
        else:
            _check(fn(self._h, _ptr(desc), len(desc), _ptr(tpl), _ptr(seq), seq.size, _ptr(refs), _ptr(out), out_stride, _ptr(skip), _ptr(off)))
        self._pending = (out, skip)
        o2, s2 = self.block_fetch()
        if inplace:
            for b in pins:
                b.free()
        return off, o2.copy(), s2.copy()

    def prepare_templates_device(self, raw, seq, misms, left_trim=(0, 0), right_trim=(0, 0), min_qual=20, keep_on_device=False,
                                 profile=None, x=None, ref=None):
        """bsc_prepare_templates_device: the read pre-processing on the GPU.  Host arrays in (uploaded with torch), the prepared
        templates / read bytes / PREP_STATS back — or, keep_on_device, the two device tensors (uint8 views) and the byte count, for
        a caller that goes on with accumulate_device / reads_chain_device."""
        import torch

        from .abi import MISMS, PREP_PARAMS, PREP_STATS, RAW_TEMPLATE

        raw = np.ascontiguousarray(raw, dtype=RAW_TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        misms = np.ascontiguousarray(misms, dtype=MISMS)
        par = np.zeros(1, dtype=PREP_PARAMS)
        par["left_trim"][0], par["right_trim"][0], par["min_qual"][0] = left_trim, right_trim, min_qual
        cap = int(seq.size) + int(np.minimum(misms["size"][misms["type"] == 1].astype(np.uint64), np.uint64(seq.size)).sum()) + 16
        dev = torch.device("cuda", torch.cuda.current_device())

        def up(a):
            b = a.view(np.uint8).reshape(-1)
            return torch.from_numpy(b.copy()).to(dev) if b.size else torch.zeros(64, dtype=torch.uint8, device=dev)

        d_raw, d_seq, d_ms = up(raw), up(seq), up(misms)
        d_tpl = torch.zeros(max(len(raw), 1) * TEMPLATE.itemsize, dtype=torch.uint8, device=dev)
        d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
        used = C.c_uint64(0)
        st = np.zeros(1, dtype=PREP_STATS)
        pf = None
        if profile is not None:  # the read profile: the block's codes (x .. y + 2) on the device, the counts on the host
            d_ref = up(np.ascontiguousarray(ref, dtype=np.uint8))
            pf = _lib.ReadProfile(d_ref.data_ptr(), int(x), len(ref), profile.counts.ctypes.data, profile.counts.shape[0], profile.used)
        _check(self._L.bsc_prepare_templates_device(self._h, d_raw.data_ptr(), len(raw), d_seq.data_ptr(), seq.size, d_ms.data_ptr(), len(misms),
                                                    _ptr(par), d_tpl.data_ptr(), d_out.data_ptr(), cap, C.byref(used), _ptr(st),
                                                    None if pf is None else C.byref(pf), None))
        if pf is not None:
            profile.used = int(pf.used)
        if keep_on_device:
            return d_tpl, d_out, int(used.value), st[0]
        tpl = d_tpl.cpu().numpy()[: len(raw) * TEMPLATE.itemsize].view(TEMPLATE).copy()
        return tpl, d_out.cpu().numpy()[: used.value].copy(), st[0]

    def block_fetch(self):
        """Wait for the submitted block and return (GT_METH[n] or uint8[n, stride], skip)."""
        if self._pending is None:
            raise BscError(-1, "block_fetch: no block was submitted")
        pending, self._pending = self._pending, None
        if not isinstance(pending[0], int):  # block_submit_to: the records are already on their way
            out, skip = pending
            _check(self._L.bsc_block_fetch(self._h, None, None))
            return out, skip
        n, stride = pending
        out = np.zeros(n, dtype=GT_METH) if stride == 200 else np.zeros((n, stride), dtype=np.uint8)
        skip = np.zeros(n, dtype=np.uint8)
        _check(self._L.bsc_block_fetch(self._h, _ptr(out), _ptr(skip)))
        return out, skip

    # -- VCF record formation (src/print_vcf.c:32-381) -------------------------------------------------
    def vcf_records(self, gtm, skip, ref, x, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None):
        """gtm: GT_METH[n] (or uint8[n, 208]) for positions x..x+n-1; ref: uint8[n+2] codes of x..x+n+1 -> VCF_CORE[n]."""
        n = len(gtm)
        stride = 200 if gtm.dtype == GT_METH else gtm.shape[1]
        gtm = np.ascontiguousarray(gtm)
        skip = np.ascontiguousarray(skip, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        if len(skip) != n or len(ref) != n + 2:
            raise ValueError("skip must have n and ref n + 2 entries")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        out = np.zeros(n, dtype=VCF_CORE)
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        _check(self._L.bsc_vcf_records(self._h, _ptr(gtm), stride, _ptr(skip), _ptr(ref), None if db is None else _ptr(db),
                                       n, x, C.byref(p), _ptr(out)))
        return out

    def vcf_records_device(self, d_gtm, stride, d_skip, d_ref, n, x, d_out, all_positions=False, reg_start=1,
                           reg_stop=0xFFFFFFFF, d_dbsnp=None, stream=None):
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        _check(self._L.bsc_vcf_records_device(self._h, d_gtm, stride, d_skip, d_ref, d_dbsnp, n, x, C.byref(p), d_out, stream))

    # -- the fused chain: pile-ups in, records (+ statistics) out, gt_meth never in HBM -------------------------------
    def chain_device(self, d_cts, d_ref, x, n_block, first, n, d_core, all_positions=False, reg_start=1,
                     reg_stop=0xFFFFFFFF, d_dbsnp=None, with_stats=False, stream=None):
        """One window (block positions first .. first + n - 1) of the block x .. x + n_block - 1.  d_cts points at the
        pile-up of block position first - min(2, first), d_ref at the reference code of first - min(4, first) (see
        include/bscall_amd.h); d_core receives n VCF_CORE records."""
        w = _lib.Window(x, n_block, first, n)
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        _check(self._L.bsc_chain_device(self._h, d_cts, d_ref, d_dbsnp, C.byref(w), C.byref(p), 1 if with_stats else 0,
                                        d_core, stream))

    def reads_chain_device(self, d_tpl, nr, d_seq, seq_bytes, x, y, d_ref, d_core, d_aux=None, all_positions=False, reg_start=1,
                           reg_stop=0xFFFFFFFF, d_dbsnp=None, with_stats=False, stream=None):
        """One block, reads in / records out in the chain's single pass (bsc_reads_chain_device): device-resident
        TEMPLATE[nr] + read bytes, d_ref = codes of x .. y + 2 -> d_core (VCF_CORE per position) [+ d_aux, 64 B per position]."""
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        _check(self._L.bsc_reads_chain_device(self._h, d_tpl, nr, d_seq, seq_bytes, x, y, d_ref, d_dbsnp, C.byref(p),
                                              1 if with_stats else 0, d_core, d_aux, stream))

    def reads_chain_len_device(self, d_tpl, nr, d_seq, seq_bytes, x, y, d_ref, d_core, d_aux, d_len, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF,
                               d_dbsnp=None, with_stats=False, stream=None):
        """bsc_reads_chain_len_device: reads_chain_device that also leaves every written record's BCF2 length in d_len (a byte per position)."""
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        _check(self._L.bsc_reads_chain_len_device(self._h, d_tpl, nr, d_seq, seq_bytes, x, y, d_ref, d_dbsnp, C.byref(p), 1 if with_stats else 0, d_core,
                                                  d_aux, d_len, stream))

    def bcf_sites_len_device(self, d_core, d_aux, d_len, n, rid, d_out, out_cap, d_totals, names=None, ids=None, stream=None):
        """bsc_bcf_sites_len_device: bcf_sites_device sized from the chain's length bytes (the records are read once)."""
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        _check(self._L.bsc_bcf_sites_len_device(self._h, d_core, d_aux, d_len, n, rid, C.byref(ids), None if nm is None else C.addressof(nm), d_out,
                                                out_cap, d_totals, stream))
        del keep

    def last_reads_chain_ms(self):
        ms = C.c_float()
        _check(self._L.bsc_last_reads_chain_ms(self._h, C.byref(ms)))
        return ms.value

    def window_size(self, limit: int) -> int:
        """The largest window <= limit positions in which every resident wave runs the same number of tiles (`bsc_chain_window_size`)."""
        return int(self._L.bsc_chain_window_size(self._h, int(limit)))

    def window_quantum(self) -> int:
        """Positions one round of the resident waves covers (`bsc_chain_window_quantum`)."""
        return int(self._L.bsc_chain_window_quantum(self._h))

    def last_chain_ms(self):
        ms = C.c_float()
        _check(self._L.bsc_last_chain_ms(self._h, C.byref(ms)))
        return ms.value

    # -- reads in, written records out (packed: only what the printer would write crosses PCIe) -----------------
    def block_records(self, templates, seq, x, y, ref, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None,
                      with_stats=False, out=None):
        """One block from reads to packed records (VCF_REC[n_written], position order).  ref: uint8[y - x + 3] codes
        of x .. y + 2.  `out`: optional preallocated VCF_REC array (capacity); default capacity = every position."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n + 2:
            raise ValueError("ref must have y - x + 3 entries (x .. y + 2)")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        if db is not None and len(db) != n:
            raise ValueError("dbsnp must have y - x + 1 entries")
        if out is None:
            out = np.zeros(n, dtype=VCF_REC)
        elif out.dtype != VCF_REC or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise ValueError("out must be a writable C-contiguous VCF_REC array")
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        cnt = C.c_uint64(0)
        _check(self._L.bsc_block_records(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(ref),
                                         None if db is None else _ptr(db), C.byref(p), 1 if with_stats else 0, _ptr(out),
                                         len(out), C.byref(cnt)))
        return out[: cnt.value]

    def block_records_raw(self, raw, seq, misms, x, y, ref, left_trim=(0, 0), right_trim=(0, 0), min_qual=20, all_positions=False,
                          reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None, with_stats=False, out=None, profile=None):
        """bsc_block_records_raw: one block from what the reader delivers (RAW_TEMPLATE[nr], read bytes, MISMS[]) to packed records;
        the read pre-processing runs on the GPU.  Returns (VCF_REC[n_written], PREP_STATS record)."""
        from .abi import MISMS, PREP_PARAMS, PREP_STATS, RAW_TEMPLATE

        raw = np.ascontiguousarray(raw, dtype=RAW_TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        misms = np.ascontiguousarray(misms, dtype=MISMS)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n + 2:
            raise ValueError("ref must have y - x + 3 entries (x .. y + 2)")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        if out is None:
            out = np.zeros(n, dtype=VCF_REC)
        par = np.zeros(1, dtype=PREP_PARAMS)
        par["left_trim"][0], par["right_trim"][0], par["min_qual"][0] = left_trim, right_trim, min_qual
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        cnt = C.c_uint64(0)
        st = np.zeros(1, dtype=PREP_STATS)
        pf = None if profile is None else _lib.ReadProfile(None, 0, 0, profile.counts.ctypes.data, profile.counts.shape[0], profile.used)
        _check(self._L.bsc_block_records_raw(self._h, _ptr(raw), len(raw), _ptr(seq), seq.size, _ptr(misms), len(misms), _ptr(par), x, y,
                                             _ptr(ref), None if db is None else _ptr(db), C.byref(p), 1 if with_stats else 0, _ptr(out),
                                             len(out), C.byref(cnt), _ptr(st), None if pf is None else C.byref(pf)))
        if pf is not None:
            profile.used = int(pf.used)
        return out[: cnt.value], st[0]

    # -- reads in, the block's BCF stream out (encoded on the device, csrc/bcfdev.hip) ---------------------------------------
    @staticmethod
    def _bcf_names(names):
        """names: None, or (pos uint32[n] ascending, off uint32[n + 1], bytes) as DbSnpIndex.names gives them -> (struct, keep-alive)."""
        if names is None:
            return None, None
        pos = np.ascontiguousarray(names[0], dtype=np.uint32)
        off = np.ascontiguousarray(names[1], dtype=np.uint32)
        by = np.frombuffer(bytes(names[2]) + b"\0", dtype=np.uint8)
        if len(off) != len(pos) + 1:
            raise ValueError("names: off must have one entry more than pos")
        st = _lib.BcfNames(pos.ctypes.data, off.ctypes.data, by.ctypes.data, len(pos))
        return st, (pos, off, by)

    def block_bcf(self, templates, seq, x, y, ref, rid, names=None, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None,
                  with_stats=False, cap=None, ids=None):
        """bsc_block_bcf: block_records with the BCF encoder behind the packing, on the device.  Returns (bytes, n_records)."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n + 2:
            raise ValueError("ref must have y - x + 3 entries (x .. y + 2)")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        if db is not None and len(db) != n:
            raise ValueError("dbsnp must have y - x + 1 entries")
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        cap_given = cap
        cap = 64 + 192 * n if cap is None else int(cap)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        nb, nr = C.c_uint64(0), C.c_uint64(0)

        def go(stats):
            return self._L.bsc_block_bcf(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(ref), None if db is None else _ptr(db),
                                         C.byref(p), stats, rid, C.byref(ids), None if nm is None else C.addressof(nm), _ptr(out), cap, C.byref(nb),
                                         C.byref(nr))

        rc = go(1 if with_stats else 0)
        if rc == -1 and nb.value > cap and cap_given is None:  # longer records than the default room: the encoder alone once more
            cap = int(nb.value)
            out = np.empty(cap, dtype=np.uint8)
            rc = self._L.bsc_block_bcf_again(self._h, _ptr(out), cap, C.byref(nb), C.byref(nr))
        _check(rc)
        del keep
        return out[: nb.value].tobytes(), nr.value

    def block_bcf_raw(self, raw, seq, misms, x, y, ref, rid, names=None, left_trim=(0, 0), right_trim=(0, 0), min_qual=20, all_positions=False,
                      reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None, with_stats=False, cap=None, profile=None, ids=None):
        """bsc_block_bcf_raw: what the reader delivers in, the block's BCF bytes out; pre-processing, calling, record formation and the
        encoding all on the device.  Returns (bytes, n_records, PREP_STATS record)."""
        from .abi import MISMS, PREP_PARAMS, PREP_STATS, RAW_TEMPLATE

        raw = np.ascontiguousarray(raw, dtype=RAW_TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        misms = np.ascontiguousarray(misms, dtype=MISMS)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n + 2:
            raise ValueError("ref must have y - x + 3 entries (x .. y + 2)")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        cap_given = cap
        cap = 64 + 192 * n if cap is None else int(cap)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        par = np.zeros(1, dtype=PREP_PARAMS)
        par["left_trim"][0], par["right_trim"][0], par["min_qual"][0] = left_trim, right_trim, min_qual
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        st = np.zeros(1, dtype=PREP_STATS)
        pf = None if profile is None else _lib.ReadProfile(None, 0, 0, profile.counts.ctypes.data, profile.counts.shape[0], profile.used)

        def go(stats, st_, pf_):
            return self._L.bsc_block_bcf_raw(self._h, _ptr(raw), len(raw), _ptr(seq), seq.size, _ptr(misms), len(misms), _ptr(par), x, y, _ptr(ref),
                                             None if db is None else _ptr(db), C.byref(p), stats, rid, C.byref(ids),
                                             None if nm is None else C.addressof(nm), _ptr(out), cap, C.byref(nb), C.byref(nr), _ptr(st_),
                                             None if pf_ is None else C.byref(pf_))

        rc = go(1 if with_stats else 0, st, pf)
        if rc == -1 and nb.value > cap and cap_given is None:  # as block_bcf
            cap = int(nb.value)
            out = np.empty(cap, dtype=np.uint8)
            rc = self._L.bsc_block_bcf_again(self._h, _ptr(out), cap, C.byref(nb), C.byref(nr))
        _check(rc)
        del keep
        if pf is not None:
            profile.used = int(pf.used)
        return out[: nb.value].tobytes(), nr.value, st[0]

    def block_bcf_rawdev(self, blk, ref, rid, names=None, left_trim=(0, 0), right_trim=(0, 0), min_qual=20, all_positions=False, reg_start=1,
                         reg_stop=0xFFFFFFFF, dbsnp=None, with_stats=False, cap=None, profile=None, ids=None):
        """bsc_block_bcf_rawdev: a block of the DEVICE reader (bamdev.DeviceBamReader.device_blocks(): raw templates, reads and lists in HBM)
        -> the block's BCF bytes; nothing of the reads crosses PCIe a second time.  Returns (bytes, n_records, PREP_STATS record)."""
        from .abi import PREP_PARAMS, PREP_STATS

        x, y = int(blk.x), int(blk.y)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = y - x + 1
        if len(ref) != n + 2:
            raise ValueError("ref must have y - x + 3 entries (x .. y + 2)")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        cap_given = cap
        cap = 64 + 192 * n if cap is None else int(cap)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        par = np.zeros(1, dtype=PREP_PARAMS)
        par["left_trim"][0], par["right_trim"][0], par["min_qual"][0] = left_trim, right_trim, min_qual
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        st = np.zeros(1, dtype=PREP_STATS)
        pf = None if profile is None else _lib.ReadProfile(None, 0, 0, profile.counts.ctypes.data, profile.counts.shape[0], profile.used)

        def go(stats, st_, pf_):
            return self._L.bsc_block_bcf_rawdev(self._h, blk.d_tpl, blk.nr, blk.d_seq, blk.seq_bytes, blk.d_misms, blk.n_misms, blk.ins_pad, _ptr(par), x, y,
                                                _ptr(ref), None if db is None else _ptr(db), C.byref(p), stats, rid, C.byref(ids),
                                                None if nm is None else C.addressof(nm), _ptr(out), cap, C.byref(nb), C.byref(nr), _ptr(st_),
                                                None if pf_ is None else C.byref(pf_))

        rc = go(1 if with_stats else 0, st, pf)
        if rc == -1 and nb.value > cap and cap_given is None:
            cap = int(nb.value)
            out = np.empty(cap, dtype=np.uint8)
            rc = self._L.bsc_block_bcf_again(self._h, _ptr(out), cap, C.byref(nb), C.byref(nr))
        _check(rc)
        del keep
        if pf is not None:
            profile.used = int(pf.used)
        return out[: nb.value].tobytes(), nr.value, st[0]

    def block_bcf_submit(self, templates, seq, x, y, ref, rid, out, names=None, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None,
                         with_stats=False, ids=None, inplace=False):
        """bsc_block_bcf_submit: queue the block and return (inputs are staged: the arrays may be reused at once); `out`: a writable uint8
        array (a PinnedBuffer's, ideally) that receives the stream; block_bcf_fetch() waits and returns (bytes view, n_records)."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n + 2:
            raise ValueError("ref must have y - x + 3 entries (x .. y + 2)")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        if db is not None and len(db) != n:
            raise ValueError("dbsnp must have y - x + 1 entries")
        if not isinstance(out, np.ndarray) or out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise ValueError("out must be a writable C-contiguous uint8 array")
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        fn = self._L.bsc_block_bcf_submit_inplace if inplace else self._L.bsc_block_bcf_submit
        _check(fn(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(ref), None if db is None else _ptr(db), C.byref(p),
                  1 if with_stats else 0, rid, C.byref(ids), None if nm is None else C.addressof(nm), _ptr(out), out.size))
        del keep
        self._bcf_out = out
        self._bcf_in = (templates, seq, ref, db) if inplace else None  # read where they lie until the fetch

    def block_bcf_fetch(self):
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        out, self._bcf_out = getattr(self, "_bcf_out", None), None
        try:
            _check(self._L.bsc_block_bcf_fetch(self._h, C.byref(nb), C.byref(nr)))
        finally:
            self._bcf_in = None
        return out[: nb.value], nr.value

    def bcf_block_device(self, d_recs, d_n_recs, max_recs, rid, d_out, out_cap, d_totals, names=None, ids=None, stream=None):
        """bsc_bcf_block_device: packed records in HBM -> their BCF stream in HBM (asynchronous on `stream`)."""
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        _check(self._L.bsc_bcf_block_device(self._h, d_recs, d_n_recs, max_recs, rid, C.byref(ids), None if nm is None else C.addressof(nm), d_out,
                                            out_cap, d_totals, stream))
        del keep

    def bcf_sites_device(self, d_core, d_aux, n, rid, d_out, out_cap, d_totals, names=None, ids=None, stream=None):
        """bsc_bcf_sites_device: the per-position arrays of reads_chain_device (d_core, d_aux) -> the BCF stream, no packing pass."""
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        _check(self._L.bsc_bcf_sites_device(self._h, d_core, d_aux, n, rid, C.byref(ids), None if nm is None else C.addressof(nm), d_out, out_cap,
                                            d_totals, stream))
        del keep

    def blocks_records(self, blocks, ref, out=None, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None, with_stats=False,
                       submit_only=False):
        """Several blocks in one launch sequence (bsc_blocks_records): blocks = [(templates, seq, x, y), ...] in genome order;
        ref = their reference codes, a list of uint8[y - x + 3] arrays (or one array, block after block); dbsnp likewise
        (y - x + 1 each) or None.  Returns (VCF_REC[n_written] of all blocks, block after block; records written per block) —
        the records of block_records() called on the blocks one after another.  submit_only: queue and return; then
        blocks_records_fetch()."""
        from .abi import BLOCK_DESC

        tpls, seqs, desc, off = [], [], np.zeros(len(blocks), dtype=BLOCK_DESC), 0
        for i, (t, sq, x, y) in enumerate(blocks):
            t = np.array(t, dtype=TEMPLATE)  # a copy: the read offsets become offsets into the joined read buffer
            sq = np.ascontiguousarray(sq, dtype=np.uint8)
            t["off"] += np.uint64(off)
            off += sq.size
            tpls.append(t)
            seqs.append(sq)
            desc[i] = (x, y, len(t), 0)
        tpl = np.concatenate(tpls) if tpls else np.zeros(0, dtype=TEMPLATE)
        seq = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
        sizes = [int(y) - int(x) + 1 for _, _, x, y in blocks]
        refs = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint8) for r in ref]) if isinstance(ref, (list, tuple)) else ref, dtype=np.uint8)
        if min(sizes) > 0 and len(refs) != sum(sizes) + 2 * len(sizes):  # (a block with y < x is the library's to refuse)
            raise ValueError("ref must hold y - x + 3 codes per block")
        db = None
        if dbsnp is not None:
            db = np.ascontiguousarray(np.concatenate([np.asarray(d, dtype=np.uint8) for d in dbsnp]) if isinstance(dbsnp, (list, tuple)) else dbsnp, dtype=np.uint8)
            if min(sizes) > 0 and len(db) != sum(sizes):
                raise ValueError("dbsnp must hold y - x + 1 flags per block")
        if out is None:
            out = np.zeros(sum(sizes), dtype=VCF_REC)
        elif out.dtype != VCF_REC or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise ValueError("out must be a writable C-contiguous VCF_REC array")
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        _check(self._L.bsc_blocks_records_submit(self._h, _ptr(desc), len(desc), _ptr(tpl), _ptr(seq), seq.size, _ptr(refs),
                                                 None if db is None else _ptr(db), C.byref(p), 1 if with_stats else 0, _ptr(out), len(out)))
        self._pending_blocks = (out, len(desc))
        return None if submit_only else self.blocks_records_fetch()

    def blocks_bcf(self, blocks, ref, rid, names=None, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None, with_stats=False, cap=None, ids=None,
                   inplace=False):
        """Several blocks in one launch sequence, their BCF bytes back as ONE stream (bsc_blocks_bcf_submit[_inplace] + _fetch): blocks, ref and
        dbsnp as blocks_records takes them; names = the table of all the blocks' flagged positions.  Returns (bytes, n_records): the streams of
        block_bcf() called on the blocks one after another, concatenated."""
        from .abi import BLOCK_DESC

        tpls, seqs, desc, off = [], [], np.zeros(len(blocks), dtype=BLOCK_DESC), 0
        for i, (t, sq, x, y) in enumerate(blocks):
            t = np.array(t, dtype=TEMPLATE)
            sq = np.ascontiguousarray(sq, dtype=np.uint8)
            t["off"] += np.uint64(off)
            off += sq.size
            tpls.append(t)
            seqs.append(sq)
            desc[i] = (x, y, len(t), 0)
        tpl = np.concatenate(tpls) if tpls else np.zeros(0, dtype=TEMPLATE)
        seq = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
        sizes = [int(y) - int(x) + 1 for _, _, x, y in blocks]
        refs = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint8) for r in ref]) if isinstance(ref, (list, tuple)) else ref, dtype=np.uint8)
        db = None
        if dbsnp is not None:
            db = np.ascontiguousarray(np.concatenate([np.asarray(d, dtype=np.uint8) for d in dbsnp]) if isinstance(dbsnp, (list, tuple)) else dbsnp, dtype=np.uint8)
        if ids is None:
            ids = _lib.BcfIds()
            self._L.bsc_bcf_default_ids(C.byref(ids))
        nm, keep = self._bcf_names(names)
        cap_given = cap
        cap = 64 + 192 * max(sum(sizes), 1) if cap is None else int(cap)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        fn = self._L.bsc_blocks_bcf_submit_inplace if inplace else self._L.bsc_blocks_bcf_submit
        _check(fn(self._h, _ptr(desc), len(desc), _ptr(tpl), _ptr(seq), seq.size, _ptr(refs), None if db is None else _ptr(db), C.byref(p),
                  1 if with_stats else 0, rid, C.byref(ids), None if nm is None else C.addressof(nm), _ptr(out), cap))
        rc = self._L.bsc_blocks_bcf_fetch(self._h, C.byref(nb), C.byref(nr))
        if rc == -1 and nb.value > cap and cap_given is None:
            cap = int(nb.value)
            out = np.empty(cap, dtype=np.uint8)
            rc = self._L.bsc_block_bcf_again(self._h, _ptr(out), cap, C.byref(nb), C.byref(nr))
        _check(rc)
        del keep, tpl, seq, refs, db, desc
        return out[: nb.value].tobytes(), nr.value

    def blocks_records_joined(self, desc, tpl, seq, ref, out, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None,
                              with_stats=False, inplace=False, submit_only=False):
        """bsc_blocks_records_submit[_inplace] on arrays that are already joined as the C ABI wants them (what a C host that
        flattens its blocks into one set of buffers holds): desc BLOCK_DESC[n_blocks], tpl TEMPLATE[sum nr] with off[] into `seq`,
        ref = the blocks' y - x + 3 codes one block after another.  inplace: no staging copy (the arrays — page-locked ones from
        PinnedBuffer make the upload a DMA — must stay unchanged until the fetch)."""
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        fn = self._L.bsc_blocks_records_submit_inplace if inplace else self._L.bsc_blocks_records_submit
        _check(fn(self._h, _ptr(desc), len(desc), _ptr(tpl), _ptr(seq), seq.size, _ptr(ref), None if dbsnp is None else _ptr(dbsnp),
                  C.byref(p), 1 if with_stats else 0, _ptr(out), len(out)))
        self._pending_blocks = (out, len(desc))
        self._pending_in = (desc, tpl, seq, ref, dbsnp)
        return None if submit_only else self.blocks_records_fetch()

    def blocks_records_fetch(self):
        out, nb = getattr(self, "_pending_blocks", None) or (None, 0)
        if out is None:
            raise BscError(-1, "blocks_records_fetch: no blocks were submitted")
        self._pending_blocks = None
        cnt, per = C.c_uint64(0), np.zeros(nb, dtype=np.uint64)
        _check(self._L.bsc_blocks_records_fetch(self._h, C.byref(cnt), _ptr(per)))
        return out[: cnt.value], per

    def block_records_submit(self, templates, seq, x, y, ref, out, all_positions=False, reg_start=1, reg_stop=0xFFFFFFFF, dbsnp=None,
                             with_stats=False, inplace=False):
        """Queue one block (reads -> packed records into `out`, a VCF_REC array that must stay alive until the fetch) and
        return at once; the inputs may be reused immediately — unless inplace (bsc_block_records_submit_inplace): then they
        are read where they lie, no staging copy, and must stay unchanged until the fetch.  block_records_fetch() completes it."""
        templates = np.ascontiguousarray(templates, dtype=TEMPLATE)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        n = int(y) - int(x) + 1
        if len(ref) != n + 2:
            raise ValueError("ref must have y - x + 3 entries (x .. y + 2)")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        if db is not None and len(db) != n:
            raise ValueError("dbsnp must have y - x + 1 entries")
        if out.dtype != VCF_REC or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise ValueError("out must be a writable C-contiguous VCF_REC array")
        p = _lib.VcfParams(1 if all_positions else 0, reg_start, reg_stop)
        fn = self._L.bsc_block_records_submit_inplace if inplace else self._L.bsc_block_records_submit
        _check(fn(self._h, _ptr(templates), len(templates), _ptr(seq), seq.size, x, y, _ptr(ref), None if db is None else _ptr(db),
                  C.byref(p), 1 if with_stats else 0, _ptr(out), len(out)))
        self._pending_rec = out
        self._pending_in = (templates, seq, ref, db) if inplace else None  # kept alive until the fetch

    def block_records_fetch(self):
        """Wait for the submitted block: VCF_REC[n_written] (a view of the array named at submit)."""
        out = getattr(self, "_pending_rec", None)
        if out is None:
            raise BscError(-1, "block_records_fetch: no block was submitted")
        self._pending_rec = None
        cnt = C.c_uint64(0)
        try:
            _check(self._L.bsc_block_records_fetch(self._h, C.byref(cnt)))
        finally:
            self._pending_in = None
        return out[: cnt.value]

    def vcf_compact_device(self, d_core, d_gtm, stride, n, d_out, out_cap, d_count, d_dbsnp=None, stream=None):
        _check(self._L.bsc_vcf_compact_device(self._h, d_core, d_gtm, stride, d_dbsnp, n, d_out, out_cap, d_count, stream))

    # -- site statistics (the sums of the reference's bs_stats; the payload of a sharded run's all-reduce) -----
    def vcf_stats(self, core, gtm, dbsnp=None):
        """Add the statistics of one block (VCF_CORE[n] from vcf_records + the GT_METH[n] behind them) to the context."""
        core = np.ascontiguousarray(core, dtype=VCF_CORE)
        gtm = np.ascontiguousarray(gtm)
        stride = gtm.dtype.itemsize if gtm.ndim == 1 else gtm.shape[1]
        if len(gtm) != len(core):
            raise ValueError("core and gtm differ in length")
        db = None if dbsnp is None else np.ascontiguousarray(dbsnp, dtype=np.uint8)
        _check(self._L.bsc_vcf_stats(self._h, _ptr(core), _ptr(gtm), stride, None if db is None else _ptr(db), len(core)))

    def vcf_stats_device(self, d_core, d_gtm, stride, n, d_dbsnp=None, stream=None):
        _check(self._L.bsc_vcf_stats_device(self._h, d_core, d_gtm, stride, d_dbsnp, n, stream))

    def site_stats(self):
        """The accumulated bsc_site_stats as a numpy record (SITE_STATS)."""
        out = np.zeros(1, dtype=SITE_STATS)
        _check(self._L.bsc_get_site_stats(self._h, _ptr(out)))
        return out[0]

    def set_gc_bins(self, d_gc, n_bins, start_pos):
        """GC bins of the contig being walked (device pointer, see `gc_bins`); None switches the GC table off."""
        _check(self._L.bsc_set_gc_bins(self._h, d_gc, n_bins, start_pos))

    def set_gc_bins_host(self, bins, start_pos):
        """The same from a host array (the context keeps its own device copy); None / empty switches the table off."""
        if bins is None or len(bins) == 0:
            _check(self._L.bsc_set_gc_bins_host(self._h, None, 0, 0))
            return
        b = np.ascontiguousarray(bins, dtype=np.uint8)
        _check(self._L.bsc_set_gc_bins_host(self._h, _ptr(b), b.size, start_pos))

    def gc_stats(self):
        """Positions by [total depth][G+C count of their 100-base bin]: the report's "GC" object, (4096, 101) uint64."""
        out = np.zeros((4096, 101), dtype=np.uint64)
        _check(self._L.bsc_get_gc_stats(self._h, _ptr(out)))
        return out

    def site_totals(self):
        """snps, indels, multi, dbSNP_sites, dbSNP_var, CpG_ref, CpG_nonref as a (7, 2) array [all, passed]: the
        reference's per-contig copy (gt_ctg_stats) is the difference of two reads."""
        out = np.zeros(14, dtype=np.uint64)
        _check(self._L.bsc_get_site_totals(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out.reshape(7, 2)

    def reset_site_stats(self):
        _check(self._L.bsc_reset_site_stats(self._h))

    # -- device-resident blocks (raw device pointers, e.g. torch tensor .data_ptr()) ---------------
    def call_sites_device(self, d_cts, d_ref, n, d_out, d_skip, out_stride=200, stream=None):
        _check(self._L.bsc_call_sites_device(self._h, d_cts, d_ref, n, d_out, out_stride, d_skip, stream))

    def synth_device(self, seed, first_site, n, coverage, d_cts, d_ref, flags=0, stream=None):
        _check(self._L.bsc_synth_pileup_device(self._h, seed, first_site, n, coverage, flags, d_cts, d_ref, stream))

    def stream_probe_ms(self, d_cts, d_ref, n, d_out, d_skip, reps=5, stream=None):
        """Best-of-reps time of the no-arithmetic copy kernel with call_sites_device's traffic (OVERWRITES d_out / d_skip)."""
        ms = C.c_float(0)
        _check(self._L.bsc_stream_probe_ms(self._h, d_cts, d_ref, n, d_out, d_skip, reps, stream, C.byref(ms)))
        return ms.value

    def set_reads_fused(self, fused=True):
        """reads -> records in ONE kernel (True) or through a pile-up in HBM (False, the default: faster)."""
        _check(self._L.bsc_set_reads_fused(self._h, 1 if fused else 0))

    def debug_fail_summary_alloc(self, on=True):
        """test hook: the two-kernel form's summaries cannot be allocated (bsc_debug_fail_summary_alloc)"""
        _check(self._L.bsc_debug_fail_summary_alloc(self._h, 1 if on else 0))

    def set_profiling(self, enable=True):
        _check(self._L.bsc_set_profiling(self._h, 1 if enable else 0))

    def last_kernel_ms(self):
        """(calling kernel ms, Fisher kernel ms) of the most recent launch, from HIP events on its stream."""
        a, b = C.c_float(), C.c_float()
        _check(self._L.bsc_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_ms_history(self, n):
        """[(calling kernel ms, Fisher kernel ms)] of the last n launches, oldest first (the library keeps the last 32):
        read after a timed loop, so that the loop itself never waits on an event."""
        out = []
        for age in range(min(int(n), 32) - 1, -1, -1):
            a, b = C.c_float(), C.c_float()
            _check(self._L.bsc_kernel_ms_history(self._h, age, C.byref(a), C.byref(b)))
            out.append((a.value, b.value))
        return out

    def synchronize(self):
        _check(self._L.bsc_synchronize(self._h))

    # -- counters ---------------------------------------------------------------------------------
    def stats(self):
        s = _lib.Stats()
        _check(self._L.bsc_get_stats(self._h, C.byref(s)))
        return {
            "sites": int(s.sites),
            "covered": int(s.covered),
            "gt_hist": [int(x) for x in s.gt_hist],
            "het_calls": int(s.het_calls),
        }

    def stats_vector(self):
        """The counters as a flat int64 vector (the block ranks all-reduce at the end of a sharded run)."""
        s = self.stats()
        return np.array([s["sites"], s["covered"]] + s["gt_hist"] + [s["het_calls"]], dtype=np.int64)

    def reset_stats(self):
        _check(self._L.bsc_reset_stats(self._h))


def synth_pileup_host(seed, first_site, n, coverage, flags=0):
    """Host twin of the device generator (bsc_synth_pileup_host): identical bits, no GPU needed."""
    L = _lib.load()
    pile = np.zeros(n, dtype=PILEUP)
    ref = np.zeros(n, dtype=np.uint8)
    _check(L.bsc_synth_pileup_host(seed, first_site, n, coverage, flags, _ptr(pile), _ptr(ref)))
    return pile, ref


def synth_reads_host(seed, x, n_sites, coverage, flags=0):
    """Synthetic read pairs over positions x .. x+n_sites-1 (bsc_synth_reads_host) -> (TEMPLATE[nt], seq uint8[])."""
    L = _lib.load()
    max_t = int(n_sites) * int(coverage) // 150 + 64
    cap = max_t * 200 + 1024
    tpl = np.zeros(max_t, dtype=TEMPLATE)
    seq = np.zeros(cap, dtype=np.uint8)
    used = C.c_uint64(0)
    nt = L.bsc_synth_reads_host(seed, x, n_sites, coverage, flags, _ptr(tpl), max_t, _ptr(seq), cap, C.byref(used))
    if nt < 0:
        raise BscError(-3, "bsc_synth_reads_host: buffer too small")
    return tpl[:nt].copy(), seq[: used.value].copy()


def synth_ref_host(seed, first_site, n, flags=0):
    """Reference codes of the synthetic genome (the ref[] of synth_pileup_host without the pile-ups)."""
    _, ref = synth_pileup_host(seed, first_site, n, 0, flags)
    return ref


class PinnedBuffer:
    """A page-locked host allocation (bsc_alloc_host) viewed as a numpy array; freed with the object."""

    def __init__(self, shape, dtype):
        self._L = _lib.load()
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if not isinstance(shape, int) else int(shape)
        self.nbytes = max(n * dt.itemsize, 1)
        self._p = self._L.bsc_alloc_host(self.nbytes)
        if not self._p:
            raise BscError(-3, self._L.bsc_last_error().decode("utf-8", "replace"))
        raw = (C.c_uint8 * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(raw, dtype=np.uint8, count=n * dt.itemsize).view(dt).reshape(shape)

    def free(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.bsc_free_host(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def gc_bins(codes):
    """ctg_stats->gc of a contig (src/read_reference.c:44-131) from its reference codes 0..4 (position 1 first):
    (start_pos, uint8 bins)."""
    L = _lib.load()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(codes.size // 100 + 1, dtype=np.uint8)
    start, nb = C.c_uint32(0), C.c_uint64(0)
    _check(L.bsc_gc_bins(_ptr(codes), codes.size, C.byref(start), _ptr(out), out.size, C.byref(nb)))
    return int(start.value), out[: nb.value].copy()


class ReadProfile:
    """bs_stats.meth_profile: the non-CpG read profile accumulated over blocks by `prepare_templates(..., profile=...)`
    (src/meth_profile.c).  `counts[i + 1]` = the four counts of read position i; `used` = the reference vector's length."""

    def __init__(self, cap=4096):
        self.counts = np.zeros((cap, 4), dtype=np.uint64)
        self.used = 0

    def reported(self):
        """What the report lists: elements 0 .. used - 1 (the report skips element 0 itself)."""
        return self.counts[: self.used]


def prepare_templates(raw, seq, misms, left_trim=(0, 0), right_trim=(0, 0), min_qual=20, profile=None, x=None, ref=None):
    """Read pre-processing on the host (bsc_prepare_templates; src/process_template.c:36-111): RAW_TEMPLATE[nr] + read
    bytes + MISMS[] -> (TEMPLATE[nr], prepared read bytes, PREP_STATS record).  No GPU involved.  With `profile` (a
    ReadProfile), the block start `x` and the block's reference codes `ref` (x .. y + 2) the templates also feed the
    non-CpG read profile."""
    from .abi import MISMS, PREP_PARAMS, PREP_STATS, RAW_TEMPLATE

    L = _lib.load()
    raw = np.ascontiguousarray(raw, dtype=RAW_TEMPLATE)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    misms = np.ascontiguousarray(misms, dtype=MISMS)
    par = np.zeros(1, dtype=PREP_PARAMS)
    par["left_trim"][0], par["right_trim"][0], par["min_qual"][0] = left_trim, right_trim, min_qual
    # room for the prepared bytes: the reads plus what their listed insertions can add; a size field larger than the block's
    # bytes is damage (the entry refuses it), not a reason to reserve gigabytes
    cap = int(seq.size) + int(np.minimum(misms["size"][misms["type"] == 1].astype(np.uint64), np.uint64(seq.size)).sum()) + 16
    out_tpl = np.zeros(len(raw), dtype=TEMPLATE)
    out_seq = np.zeros(cap, dtype=np.uint8)
    used = C.c_uint64(0)
    st = np.zeros(1, dtype=PREP_STATS)
    if profile is None:
        _check(L.bsc_prepare_templates(_ptr(raw), len(raw), _ptr(seq), seq.size, _ptr(misms), len(misms), _ptr(par), _ptr(out_tpl),
                                       _ptr(out_seq), cap, C.byref(used), _ptr(st)))
    else:
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        pf = _lib.ReadProfile(ref.ctypes.data, int(x), ref.size, profile.counts.ctypes.data, profile.counts.shape[0], profile.used)
        _check(L.bsc_prepare_templates_profile(_ptr(raw), len(raw), _ptr(seq), seq.size, _ptr(misms), len(misms), _ptr(par),
                                               _ptr(out_tpl), _ptr(out_seq), cap, C.byref(used), _ptr(st), C.byref(pf)))
        profile.used = int(pf.used)
    return out_tpl, out_seq[: used.value].copy(), st[0]
