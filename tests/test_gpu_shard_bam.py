"""File-fed sharding, the rehearsal on ONE GPU: two ranks (both on cuda:0, gloo between them) of pipeline.run(shard_rank=..., shard_world=2)
over one multi-contig BAM — every rank reads the stretches of the file that hold its contigs (the device reader's contig selection), calls
them, writes its contigs' shards; the ranks all-reduce what the report sums; rank 0 concatenates the shards in contig order.  The BCF and
the report are, byte for byte, the single run's."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_bam_s", os.path.join(ROOT, "tools", "make_bam.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)


def _reference():
    rng = np.random.default_rng(123)
    ref = {"chrA": rng.integers(1, 5, 30_000).astype(np.uint8), "chrB": rng.integers(1, 5, 45_000).astype(np.uint8),
           "chrC": rng.integers(1, 5, 12_000).astype(np.uint8), "chrD": rng.integers(1, 5, 20_000).astype(np.uint8)}
    ref["chrB"][7_000:7_300] = 0
    return ref


def _worker(rank, world, port, bam, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from bs_call_amd import pipeline

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipeline.run(bam, _reference(), os.path.join(out_dir, "sharded.bcf"), sample="S1", report_path=os.path.join(out_dir, "sharded.json"), date=(3, 10, 2026),
                 compressed=False, shard_rank=rank, shard_world=world)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_write_the_single_runs_bytes(tmp_path):
    import torch.multiprocessing as mp

    from bs_call_amd import pipeline

    reference = _reference()
    refs = [(k, len(v)) for k, v in reference.items()]
    rng = np.random.default_rng(9)
    recs = []
    for tid, (name, n) in enumerate(refs):
        if name == "chrC":
            continue  # a contig without reads
        recs += W.wgbs_records(rng, reference[name], tid, n * 30 // 200, het_every=500)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bam = str(tmp_path / "in.bam")
    W.write_bam(bam, refs, recs, aligned=True)
    single = pipeline.run(bam, reference, str(tmp_path / "single.bcf"), sample="S1", report_path=str(tmp_path / "single.json"), date=(3, 10, 2026),
                          compressed=False, device_reader=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, bam, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "sharded.bcf", "rb").read() == open(tmp_path / "single.bcf", "rb").read()
    assert open(tmp_path / "sharded.json").read() == open(tmp_path / "single.json").read()
    assert single["records"] > 20_000 and single["contigs"] == ["chrA", "chrB", "chrD"]
    assert not [f for f in os.listdir(tmp_path) if ".shard" in f]  # the shards are gone


def test_plain_c_ranks_and_merge_write_the_single_runs_bytes(tmp_path):
    """integration/bam2bcf --rank r --world n / --merge n (tools/bam2bcf_sharded.sh): the C twin of the sharded pipeline — ranks that never
    talk to each other, their shards and sums on disk, one merge: the single run's BCF and report bytes; for 2 and for 3 ranks (one of them
    gets the contig without reads, or nothing but it)."""
    import subprocess

    exe = os.path.join(ROOT, "bs_call_amd", "lib", "bam2bcf")
    assert os.path.exists(exe), "run `make demo`"
    reference = _reference()
    refs = [(k, len(v)) for k, v in reference.items()]
    rng = np.random.default_rng(19)
    recs = []
    for tid, (name, n) in enumerate(refs):
        if name == "chrC":
            continue
        recs += W.wgbs_records(rng, reference[name], tid, n * 20 // 200, het_every=700)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bam, fa = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    W.write_bam(bam, refs, recs, aligned=True)
    with open(fa, "w") as f:
        for name, codes in reference.items():
            f.write(">%s\n" % name)
            s = "".join("NACGT"[c] for c in codes)
            for o in range(0, len(s), 70):
                f.write(s[o : o + 70] + "\n")
    r = subprocess.run([exe, bam, fa, str(tmp_path / "one.bcf"), str(tmp_path / "one.json"), "S3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    want = (open(tmp_path / "one.bcf", "rb").read(), open(tmp_path / "one.json").read(), r.stdout.strip())
    assert len(want[0]) > 1_000_000
    for world in (2, 3):
        out, rep = str(tmp_path / ("w%d.bcf" % world)), str(tmp_path / ("w%d.json" % world))
        r = subprocess.run([os.path.join(ROOT, "tools", "bam2bcf_sharded.sh"), str(world), bam, fa, out, rep, "S3"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, BAM2BCF_ONE_GPU="1"))
        assert r.returncode == 0, r.stderr + r.stdout
        assert open(out, "rb").read() == want[0], world
        assert open(rep).read() == want[1], world
        assert r.stdout.strip().splitlines()[-1] == want[2]
        assert not [f for f in os.listdir(tmp_path) if ".shard" in f or ".sums" in f]
    # the same records in a file whose BGZF blocks cut records (htsjdk's writer): the ranks find the record starts themselves — same bytes
    cut = str(tmp_path / "cut.bam")
    W.write_bam(cut, refs, recs, aligned=False, block=3000)
    out, rep = str(tmp_path / "cut.bcf"), str(tmp_path / "cut.json")
    r = subprocess.run([os.path.join(ROOT, "tools", "bam2bcf_sharded.sh"), "3", cut, fa, out, rep, "S3"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, BAM2BCF_ONE_GPU="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    assert open(out, "rb").read() == want[0] and open(rep).read() == want[1]
