"""py_dbsnp.py — pure-Python restatement of the reference's dbSNP index reader (src/dbSNP.c:27-350), the checker of the
library's C reader (csrc/dbsnp.c).  *** TEST INFRASTRUCTURE, NOT PRODUCT ***.  Follows the reference's statements
(load_dbSNP_header, load_dbSNP_ctg, dbSNP_lookup_name) with Python's own struct / zlib; small files only."""
import struct
import zlib

MAGIC = 0xD7278434
# db_tab (src/dbSNP.c:160-177) as its rule: 0x21 + n -> BCD of n (00..99); 0x85 + d -> d, filler
DB_TAB = [0xFF] * 256
for _n in range(100):
    DB_TAB[0x21 + _n] = ((_n // 10) << 4) | (_n % 10)
for _d in range(10):
    DB_TAB[0x85 + _d] = (_d << 4) | 0xF


class Index:
    def __init__(self, path):
        self.data = open(path, "rb").read()
        magic, _res, off, bufsize, csz = struct.unpack_from("<IIQQQ", self.data, 0)
        if magic != MAGIC:
            raise ValueError("Invalid format")
        (tail,) = struct.unpack_from("<I", self.data, off + csz)
        if tail != MAGIC:
            raise ValueError("directory not followed by the magic")
        d = zlib.decompress(self.data[off : off + csz])
        self.bufsize = bufsize
        (self.n_prefixes,) = struct.unpack_from("<H", d, 2)
        (n_ctgs,) = struct.unpack_from("<I", d, 4)
        p = 8
        e = d.index(b"\0", p)
        if not d[p:e].startswith(b"track "):
            raise ValueError("no track header")
        self.header = d[p + 6 : e].decode()
        p = e + 1
        self.prefixes = []
        for _ in range(self.n_prefixes):
            e = d.index(b"\0", p)
            self.prefixes.append(d[p:e].decode())
            p = e + 1
        self.ctgs = {}
        for _ in range(n_ctgs):
            min_bin, max_bin, offset = struct.unpack_from("<IIQ", d, p)
            p += 16
            e = d.index(b"\0", p)
            self.ctgs[d[p:e].decode()] = (min_bin, max_bin, offset)
            p = e + 1
        self.bins = None

    def load_contig(self, name):
        """src/dbSNP.c:157-304 -> {bin number: (mask, fq_mask, entries, name_buf)}"""
        self.bins = {}
        if name not in self.ctgs:
            return 0
        min_bin, max_bin, off = self.ctgs[name]
        self.range = (min_bin, max_bin)
        curr_bin, n_snps = min_bin, 0
        while True:
            (sz,) = struct.unpack_from("<Q", self.data, off)
            off += 8
            if sz == 0:
                break
            buf = zlib.decompress(self.data[off : off + sz])
            off += sz
            bp, end = 0, len(buf)
            entries, name_buf, mask, fq, prev_ix = [], bytearray(), 0, 0, -1
            while bp < end:
                if not entries:
                    x = buf[bp]
                    bp += 1
                    t = x & 3
                    if t == 0:
                        inc = x >> 2
                    elif t == 1:
                        inc = buf[bp]
                        bp += 1
                    elif t == 2:
                        (inc,) = struct.unpack_from("<H", buf, bp)
                        bp += 2
                    else:
                        (inc,) = struct.unpack_from("<I", buf, bp)
                        bp += 4
                    curr_bin += inc
                    if curr_bin > max_bin or bp >= end:
                        break
                x = buf[bp]
                bp += 1
                prefix_ix = x >> 6
                if not prefix_ix:
                    name_buf += buf[bp : bp + 2]
                    bp += 2
                if (x & 63) <= prev_ix or prefix_ix > self.n_prefixes:
                    raise ValueError("entries out of order")
                prev_ix = x & 63
                k = 0
                while bp < end and buf[bp] > 3:
                    name_buf.append(DB_TAB[buf[bp]])
                    bp += 1
                    k += 1
                tm = buf[bp]
                bp += 1
                mask |= 1 << prev_ix
                if tm & 2:
                    fq |= 1 << prev_ix
                entries.append((k << 8) | x)
                if tm & 1:
                    self.bins[curr_bin] = (mask, fq, entries, bytes(name_buf))
                    n_snps += len(entries)
                    entries, name_buf, mask, fq, prev_ix = [], bytearray(), 0, 0, -1
        return n_snps

    def lookup(self, x):
        """dbSNP_lookup_name (src/dbSNP.c:306-350) -> (rs_found, name, rs_len)"""
        dtab = "0123456789" + "\0" * 6
        b = self.bins.get(x >> 6) if self.bins else None
        if b is None:
            return 0, "", 0
        mask, fq, entries, name_buf = b
        mk = 1 << (x & 63)
        if not (mask & mk):
            return 0, "", 0
        res = 3 if fq & mk else 1
        mk1 = mask & (mk - 1)
        i = j = 0
        while mk1:
            if mk1 & 1:
                en = entries[i]
                i += 1
                j += en >> 8
                if not ((en >> 6) & 3):
                    j += 2
            mk1 >>= 1
        prefix_id = (entries[i] >> 6) & 3
        tp1 = j
        if prefix_id == 0:
            prefix_id = (name_buf[tp1] << 8) | name_buf[tp1 + 1]
            tp1 += 2
        else:
            prefix_id -= 1
        rs = self.prefixes[prefix_id]
        for k in range(entries[i] >> 8):
            z = name_buf[tp1 + k]
            rs += dtab[z >> 4] + dtab[z & 15]
        return res, rs.split("\0")[0], len(rs)
