#!/usr/bin/env python3
"""Writer of bs_call's compressed dbSNP index — the on-disk format bin/dbSNP_idx produces (reference
src/dbSNP_output.c:139-182 finish_output, :202-299 output_contig; read back by src/dbSNP.c and by csrc/dbsnp.c).

  file    = 32-byte header {u32 magic 0xd7278434, u32 0, u64 directory offset, u64 largest uncompressed block,
            u64 compressed directory size}, contig data, zlib(directory), u32 magic
  contig  = blocks {u64 compressed size, zlib(block)} of at most 2 048 bins each, then u64 0
  block   = per non-empty bin (64 positions: bin = x >> 6): distance from the previous bin (k < 64: one byte k << 2;
            k < 256: 1, k; k < 65 536: 2, u16; else 3, u32), then its entries in position order:
            {(x & 63) | (prefix index + 1) << 6   [prefix index >= 3: top bits 0, then the index as u16],
             name digits (two per byte, 0x21 + value; a last odd digit d as 0x85 + d), terminator: bit 0 = last entry of
             the bin, bit 1 = fq_mask (the site's homozygous-reference record is forced out, src/print_vcf.c:139)}
  directory = {u8 version 2, u8 0, u16 n_prefixes, u32 n_contigs, "track ..." header \\0, prefixes \\0, per contig
              {u32 min_bin, u32 max_bin, u64 file offset, name \\0}}

As a module: write_index(path, contigs, prefixes=("rs",)), contigs = {name: [(position, rs digits string, fq flag,
prefix index), ...]}.  As a script: the SURVEY.md 8(d) synthetic index (one site per `--spacing` bp, 10 % fq_mask):
  python tools/make_dbsnp_index.py out.idx --contig chrS:1000000 [--contig name:length ...] [--spacing 300]
"""
import struct
import sys
import zlib

MAGIC = 0xD7278434
ITEMS_PER_BLOCK = 2048  # include/dbSNP_idx.h:26


def _digits(rs):
    out = bytearray()
    for i in range(0, len(rs) - 1, 2):
        out.append(0x21 + int(rs[i]) * 10 + int(rs[i + 1]))
    if len(rs) & 1:
        out.append(0x85 + int(rs[-1]))
    return bytes(out)


def write_index(path, contigs, prefixes=("rs",), header="track name = dbSNP_index description = \"dbSNP index produced by dbSNP_idx\""):
    max_buf = 0
    directory = []
    with open(path, "wb") as fp:
        fp.write(b"\0" * 32)
        for name, sites in contigs.items():
            bins = {}
            for pos, rs, fq, pix in sites:
                bins.setdefault(pos >> 6, []).append((pos & 63, rs, fq, pix))
            if not bins:
                continue
            min_bin, max_bin = min(bins), max(bins)
            offset = fp.tell()
            buf, n_items, curr = bytearray(), 0, min_bin
            for b in sorted(bins):
                k = b - curr
                if k < 64:
                    buf.append(k << 2)
                elif k < 256:
                    buf += bytes([1, k])
                elif k < 65536:
                    buf += bytes([2]) + struct.pack("<H", k)
                else:
                    buf += bytes([3]) + struct.pack("<I", k)
                curr = b
                ents = sorted(bins[b])
                assert len(ents) <= 64 and len({e[0] for e in ents}) == len(ents), "one entry per position"
                for j, (ix, rs, fq, pix) in enumerate(ents):
                    if pix < 3:
                        buf.append(ix | ((pix + 1) << 6))
                    else:
                        buf.append(ix)
                        buf += struct.pack("<H", pix)
                    buf += _digits(rs)
                    buf.append((2 if fq else 0) | (1 if j == len(ents) - 1 else 0))
                n_items += 1
                if n_items == ITEMS_PER_BLOCK:
                    max_buf = max(max_buf, len(buf))
                    z = zlib.compress(bytes(buf))
                    fp.write(struct.pack("<Q", len(z)) + z)
                    buf, n_items = bytearray(), 0
            if n_items:
                max_buf = max(max_buf, len(buf))
                z = zlib.compress(bytes(buf))
                fp.write(struct.pack("<Q", len(z)) + z)
            fp.write(struct.pack("<Q", 0))
            directory.append((min_bin, max_bin, offset, name))
        off = fp.tell()
        d = bytearray(struct.pack("<BBHI", 2, 0, len(prefixes), len(directory)))
        d += header.encode() + b"\0"
        for p in prefixes:
            d += p.encode() + b"\0"
        for min_bin, max_bin, offset, name in directory:
            d += struct.pack("<IIQ", min_bin, max_bin, offset) + name.encode() + b"\0"
        max_buf = max(max_buf, len(d))
        z = zlib.compress(bytes(d))
        fp.write(z)
        fp.write(struct.pack("<I", MAGIC))
        fp.seek(0)
        fp.write(struct.pack("<IIQQQ", MAGIC, 0, off, max_buf, len(z)))


def synthetic_sites(length, spacing=300, seed=88172645463325252 + 5, first_rs=1000):
    """one site per `spacing` bp (jittered inside its stretch), 10 % flagged fq_mask; xorshift64."""
    x = seed & 0xFFFFFFFFFFFFFFFF
    sites, rs = [], first_rs
    for start in range(1, length + 1, spacing):
        x ^= (x << 13) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 7
        x ^= (x << 17) & 0xFFFFFFFFFFFFFFFF
        pos = start + x % min(spacing, length - start + 1)
        sites.append((pos, str(rs), (x >> 32) % 10 == 0, 0))
        rs += 1 + (x >> 40) % 7
    return sites


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--contig", action="append", default=[], help="name:length")
    ap.add_argument("--spacing", type=int, default=300)
    a = ap.parse_args()
    ctgs, first = {}, 1000
    for c in a.contig or ["chrS:1000000"]:
        name, length = c.rsplit(":", 1)
        ctgs[name] = synthetic_sites(int(length), a.spacing, first_rs=first)
        first += 10 * len(ctgs[name])
    write_index(a.out, ctgs)
    print("%s: %d contigs, %d sites" % (a.out, len(ctgs), sum(len(v) for v in ctgs.values())), file=sys.stderr)
