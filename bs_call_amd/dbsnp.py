"""ctypes mirror of the dbSNP index reader (csrc/dbsnp.c; include/bscall_amd.h): load_dbSNP_header / load_dbSNP_ctg /
dbSNP_lookup_name of the reference (src/dbSNP.c), plus the per-position flags the device entry points take."""
import ctypes as C

import numpy as np

from . import _lib
from .caller import BscError


def _check(rc):
    if rc < 0:
        raise BscError(rc, _lib.load().bsc_last_error().decode("utf-8", "replace"))
    return rc


class DbSnpIndex:
    def __init__(self, path):
        self._L = _lib.load()
        h = C.c_void_p()
        _check(self._L.bsc_dbsnp_open(str(path).encode(), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.bsc_dbsnp_close(self._h)
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

    @property
    def contigs(self):
        return [self._L.bsc_dbsnp_contig_name(self._h, i).decode("utf-8", "replace") for i in range(self._L.bsc_dbsnp_n_contigs(self._h))]

    @property
    def header(self):
        return self._L.bsc_dbsnp_header(self._h).decode("utf-8", "replace")

    def load_contig(self, name):
        """Make `name` the loaded contig (the previous one is dropped); returns the number of entries (0 for a contig the
        index does not list)."""
        n = C.c_uint64(0)
        _check(self._L.bsc_dbsnp_load_contig(self._h, name.encode(), C.byref(n)))
        return n.value

    def flags(self, x0, n):
        """rs_found (0 / 1 / 3) of positions x0 .. x0 + n - 1 (1-based) of the loaded contig, uint8[n]."""
        out = np.zeros(n, dtype=np.uint8)
        _check(self._L.bsc_dbsnp_flags(self._h, x0, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def names(self, x0, n):
        """The names of the flagged positions of x0 .. x0 + n - 1 of the loaded contig (bsc_dbsnp_names): (pos uint32[k] ascending,
        off uint32[k + 1], bytes) — the table SiteCaller.block_bcf* hands to the device encoder."""
        k, nb = C.c_uint32(0), C.c_uint64(0)
        _check(self._L.bsc_dbsnp_names(self._h, x0, n, None, None, None, 0, 0, C.byref(k), C.byref(nb)))
        pos = np.zeros(k.value, dtype=np.uint32)
        off = np.zeros(k.value + 1, dtype=np.uint32)
        by = np.zeros(nb.value + 1, dtype=np.uint8)
        _check(self._L.bsc_dbsnp_names(self._h, x0, n, pos.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), by.ctypes.data_as(C.c_void_p),
                                       k.value, nb.value, C.byref(k), C.byref(nb)))
        return pos, off, by[: nb.value].tobytes()

    def name(self, x):
        """(rs_found, name, rs_len as the reference counts it) of position x."""
        buf = C.create_string_buffer(600)
        ln = C.c_size_t(0)
        r = _check(self._L.bsc_dbsnp_name(self._h, x, buf, 600, C.byref(ln)))
        return r, buf.value.decode("utf-8", "replace"), ln.value
