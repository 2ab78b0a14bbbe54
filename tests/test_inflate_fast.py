"""csrc/inflate_fast.c (the device reader's host half inflates BGZF blocks with it) against zlib: every level and strategy over data of every
kind — random bytes (stored blocks, literal-heavy dynamic codes), text, long runs (distance 1 .. 7 matches), BAM-like records —, fixed-Huffman
and stored streams, block sizes 0 .. 65 536, and for every stream its truncations and bit flips: the routine must give zlib's bytes where zlib
accepts, and refuse without reading or writing out of bounds (the buffers carry guard bytes) what zlib refuses."""
import ctypes as C
import zlib

import numpy as np
import pytest

from bs_call_amd import _lib


@pytest.fixture(scope="module")
def L():
    return _lib.load()


def inflate(L, comp, n_out, guard=64):
    out = (C.c_uint8 * (n_out + guard))()
    C.memset(out, 0xA5, n_out + guard)
    src = (C.c_uint8 * (len(comp) + 1)).from_buffer_copy(bytes(comp) + b"\0")
    rc = L.bsc_inflate_raw(src, len(comp), out, n_out)
    assert bytes(out[n_out:]) == b"\xa5" * guard, "wrote behind the output"
    return rc, bytes(out[:n_out])


def datasets(rng):
    yield b""
    yield b"a"
    yield bytes(rng.integers(0, 256, 70, dtype=np.uint8))
    yield bytes(rng.integers(0, 256, 65536, dtype=np.uint8))             # incompressible: stored or near-uniform codes
    yield bytes(rng.integers(0, 4, 65536, dtype=np.uint8))                # 2-bit entropy: short codes, many matches
    yield bytes(rng.integers(20, 44, 60000, dtype=np.uint8))              # base qualities
    yield b"ab" * 30000                                                   # distance 2
    yield b"x" * 65536                                                    # distance 1, longest matches
    yield (b"abcdefg" * 9000)[:60001]                                     # distance 7
    yield b"".join(b"t%09d\0" % i + bytes(rng.integers(0, 256, 50, dtype=np.uint8)) + bytes(rng.integers(20, 44, 100, dtype=np.uint8)) for i in range(380))
    text = (b"The quick brown fox jumps over the lazy dog. " * 40 + bytes(rng.integers(0, 256, 200, dtype=np.uint8))) * 30
    yield text[:65536]
    # a skewed alphabet: code lengths up to 15 bits (second-level tables)
    p = np.array([2.0 ** -min(i // 4 + 1, 40) for i in range(256)])
    yield bytes(rng.choice(256, 65536, p=p / p.sum()).astype(np.uint8))


def test_equals_zlib_on_every_level_and_strategy(L):
    rng = np.random.default_rng(1)
    n = 0
    for data in datasets(rng):
        for level in (0, 1, 2, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                comp = co.compress(data) + co.flush()
                rc, out = inflate(L, comp, len(data))
                assert rc == 0 and out == data, (len(data), level, strategy)
                assert L.bsc_crc32(data, len(data)) == zlib.crc32(data)
                n += 1
                if len(data) > 0:  # the right stream into the wrong size: refused
                    assert inflate(L, comp, len(data) - 1)[0] == -1
                    assert inflate(L, comp, len(data) + 1)[0] == -1
    assert n > 300


def test_damaged_streams_are_refused_or_equal_zlib(L):
    rng = np.random.default_rng(2)
    checked = refused = 0
    for data in list(datasets(rng))[3:]:
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        comp = bytearray(co.compress(data) + co.flush())
        for trial in range(60):
            bad = bytearray(comp)
            kind = trial % 3
            if kind == 0:
                bad = bad[: int(rng.integers(0, len(bad)))]
            elif kind == 1:
                k = int(rng.integers(0, len(bad)))
                bad[k] ^= 1 << int(rng.integers(0, 8))
            else:
                k = int(rng.integers(0, max(1, len(bad) - 8)))
                bad[k : k + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
            try:
                d = zlib.decompressobj(-15)
                want = d.decompress(bytes(bad)) + d.flush()
                ok = d.eof and len(want) == len(data)
            except zlib.error:
                ok = False
            rc, out = inflate(L, bad, len(data))
            checked += 1
            if ok:
                assert rc == 0 and out == want
            else:
                assert rc == -1
                refused += 1
    assert checked > 400 and refused > 200


def test_crc_of_odd_pieces(L):
    rng = np.random.default_rng(3)
    buf = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))
    for a, b in ((0, 0), (0, 1), (1, 2), (3, 20), (5, 4099), (7, 5000), (0, 5000)):
        piece = buf[a:b]
        assert L.bsc_crc32(piece, len(piece)) == zlib.crc32(piece)
