"""The N > 1 code path of bench.py on the one-GPU box: the bare `python bench.py --gpus 2 --config 3 --genome-scale 0.004`
(bench.py starts its own ranks: launch_ranks -> torch.distributed.run, the parent never touches the GPU) with BENCH_REHEARSAL=1 (both ranks on cuda:0, gloo instead of RCCL,
which refuses two ranks on one device) — contig partition (the reference's unit: one process per contig, README.md:73-76),
the all-reduce of the counter and statistics blocks and the gather of the per-contig totals, against the N = 1 run of the
same genome.  The numbers of a rehearsal mean nothing; the sums must be the same."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env=None):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


def test_two_rank_rehearsal_equals_single_process():
    common = ["bench.py", "--config", "3", "--genome-scale", "0.004", "--steps", "2", "--warmup", "1"]
    one = _run([sys.executable] + common + ["--gpus", "1"])
    env = dict(os.environ, BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # exactly what the driver types for N = 2, no external launcher
    two = _run([sys.executable] + common + ["--gpus", "2"], env)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    c1, c2 = one["config"], two["config"]
    assert c2["per_contig_records_sum_equals_total"] is True and c1["per_contig_records_sum_equals_total"] is True
    assert c2["contigs_with_records_after_gather"] == 24 == c1["contigs_with_records_after_gather"]
    for k in ("records_written", "CpGs", "dbSNP_sites_written", "covered_fraction"):
        assert c1[k] == c2[k], (k, c1[k], c2[k])
    assert "rank 0 of 2" in c2["share"]
    # the N > 1 line explains itself: strong scaling, who owned what (whole contigs, longest-processing-time), max / mean rank time
    shares = c2["shares"]
    assert [sh["rank"] for sh in shares] == [0, 1] and sorted(c for sh in shares for c in sh["contigs"]) == list(range(24))
    assert sum(sh["positions"] for sh in shares) == c1["shares"][0]["positions"] and max(sh["of_largest"] for sh in shares) == 1.0
    assert min(sh["of_largest"] for sh in shares) > 0.9  # LPT over 24 human-length contigs balances two ranks within a few percent
    rt = two["rank_time"]
    assert len(rt["per_rank_s"]) == 2 and abs(rt["max_s"] - max(rt["per_rank_s"])) < 1e-5 and rt["max_over_mean"] >= 1.0
    assert abs(two["ms_per_step"] * two["steps"] / 1e3 - rt["max_s"]) < 1e-3  # the value is positions / the slowest rank's time
    assert one["rank_time"]["max_over_mean"] == 1.0 and len(c1["shares"]) == 1
    # configs[1] (the default line): weak scaling, one contig per rank, the same summary
    w2 = _run([sys.executable, "bench.py", "--gpus", "2", "--sites", "2000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-chain", "--no-reads"], env)
    assert w2["n_gpus"] == 2 and w2["scaling"] == "weak" and len(w2["rank_time"]["per_rank_s"]) == 2 and w2["rank_time"]["max_over_mean"] >= 1.0


def test_external_launcher_still_works():
    """The driver's other form: torch.distributed.run around bench.py (WORLD_SIZE set => bench.py does not launch again)."""
    common = ["bench.py", "--config", "3", "--genome-scale", "0.002", "--steps", "1", "--warmup", "1"]
    env = dict(os.environ, BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())] + common + ["--gpus", "2"], env)
    assert two["n_gpus"] == 2 and two["config"]["per_contig_records_sum_equals_total"] is True
