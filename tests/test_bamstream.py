"""csrc/bamstream.c — the host half of the device reader, on the CPU (without a device its slabs are ordinary memory): the inflated
bytes are the file's, the record offsets are the block_size chain's, whatever the BGZF block size, slab size and helper count — records
that straddle blocks and slabs, size fields cut in two, empty blocks in mid-file — and damaged files fail without a hang."""
import gzip
import importlib.util
import os
import struct

import numpy as np
import pytest

from bs_call_amd.bamdev import BamStream
from bs_call_amd.caller import BscError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("test_bam_mod2", os.path.join(ROOT, "tests", "test_bam.py"))
TB = importlib.util.module_from_spec(spec)
spec.loader.exec_module(TB)
W = TB.W


def expected(path):
    raw = gzip.open(path, "rb").read()
    o = 8 + struct.unpack_from("<I", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    for _ in range(n_ref):
        o += 8 + struct.unpack_from("<I", raw, o)[0]
    first = o
    offs = []
    while o < len(raw):
        offs.append(o)
        o += 4 + struct.unpack_from("<I", raw, o)[0]
    return raw, first, offs


def streamed(path, **kw):
    data, offs = bytearray(), []
    with BamStream(path, **kw) as s:
        refs, text, first, nth = s.refs, s.header_text, s.first_record, s.threads
        last_seen = False
        for off, b, ro, last in s.slabs():
            assert not last_seen and off == len(data)
            data += b
            offs += [off + int(v) for v in ro]
            last_seen = last
        assert last_seen
    return bytes(data), offs, refs, text, first, nth


@pytest.mark.parametrize("block,slab,threads", [(0xFF00, 0, 1), (777, 65536, 3), (4096, 65536, 8), (100, 65536, 2), (0xFF00, 70000, 5)])
def test_stream_equals_the_inflated_file(tmp_path, block, slab, threads):
    rng = np.random.default_rng(block + threads)
    recs = TB._random_records(rng, 3000)
    p = str(tmp_path / "s.bam")
    W.write_bam(p, TB.REFS, recs, block=block)
    raw, first, offs = expected(p)
    data, got, refs, text, first_got, nth = streamed(p, threads=threads, slab_bytes=slab, n_slabs=3)
    assert data == raw and got == offs and first_got == first and refs == TB.REFS and text.startswith("@HD") and nth == threads


def test_empty_blocks_header_only_and_damage(tmp_path):
    p = str(tmp_path / "e.bam")
    # an end-of-file marker in mid-file (concatenated BAM pieces) is stepped over
    recs = [TB.rec("a", 0, 100, -1), TB.rec("b", 0, 200, -1)]
    W.write_bam(p, TB.REFS, recs, block=120)
    raw = open(p, "rb").read()
    cut = raw.index(W.BGZF_EOF)
    open(p, "wb").write(raw[:cut] + W.BGZF_EOF + W.BGZF_EOF)
    want, first, offs = expected(p)
    data, got, _, _, _, _ = streamed(p, threads=2, slab_bytes=65536)
    assert data == want and got == offs and len(offs) == 2
    # a header and nothing else
    W.write_bam(p, TB.REFS, [])
    data, got, refs, _, first, _ = streamed(p, threads=2)
    assert got == [] and first == len(data) and refs == TB.REFS
    # truncated in a record / in a block / a flipped payload byte / not a BAM file: an error, never a hang
    W.write_bam(p, TB.REFS, TB._random_records(np.random.default_rng(5), 500), block=3000)
    raw = open(p, "rb").read()
    bad = str(tmp_path / "bad.bam")
    for damage in (raw[: len(raw) // 2], raw[:-40], raw[:200] + bytes([raw[200] ^ 0x55]) + raw[201:], b"\x1f\x8b\x08\x00" + b"\0" * 30, b"hello"):
        open(bad, "wb").write(damage)
        with pytest.raises(BscError):
            streamed(bad, threads=3, slab_bytes=65536)
    # a record cut off at the end of the last block: "truncated"
    want, first, offs = expected(p)
    body = want[: offs[-1] + 10]
    with open(bad, "wb") as f:
        for o in range(0, len(body), 5000):
            f.write(W.bgzf_block(body[o : o + 5000]))
        f.write(W.BGZF_EOF)
    with pytest.raises(BscError, match="truncated"):
        streamed(bad, threads=2)
    # closed half way, helpers still busy: no hang
    s = BamStream(p, threads=4, slab_bytes=65536, n_slabs=2)
    it = s.slabs()
    next(it)
    s.close()
