#!/usr/bin/env python3
"""bench.py — genome positions called per second by the gfx950 calling path.

Contract (see the task prompt):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.

Default workload = BASELINE.json configs[1] (synthetic chr22-sized contig, 50 Mb at 30x WGBS):
  * a "step" = one pass of the hot path (pile-up -> gt_meth: bsc_call_kernel + the Fisher pass) over this rank's contig,
    pile-ups and reference codes already resident in HBM (bsc_synth_pileup_device); once at the end the RCCL all-reduce
    of the per-rank counters.  N > 1: one process per GPU, every rank its own contig, no data-path collective (weak
    scaling); value = all positions / max-over-ranks time.
  * roofline: achieved = 305 algorithmic bytes per covered position (104 B pile-up + 1 B reference code + 200 B gt_meth,
    SURVEY.md 8d; 105 B per uncovered one) x positions per launch / average device time of bsc_call_kernel, HIP events
    recorded on the launch stream inside libbscall_amd (bsc_set_profiling / bsc_last_kernel_ms).
  * roofline_chain: the same for the fused print-side chain (bsc_chain_device: pile-up -> call -> VCF record -> site
    statistics, csrc/fused.hip), 105 B in + 64 B out per position — measured after the timed region.
  * roofline_accumulate / roofline_reads (N = 1): HOT LOOP A (reference src/call_genotypes.c:180-226) and the reads-in chain
    (reads -> records in one kernel, neither pile-up nor gt_meth in HBM) over device-resident L-reads of the same contig
    (1 B per base + 16 B per template in; 104 B pile-up, or 1 B reference code in + 64 B record out, per
    position), device time from HIP events around all launches of the stage — measured after the timed region.
  * roofline_bcf (N = 1): the tail of record formation on the device (csrc/bcfdev.hip) over the packed written records of the reads
    leg's block: 2 x 128 B read + the BCF bytes written per record, events on the launch stream; first records = the host encoder's.
  * every leg also carries `valu`: the fraction of the VALU issue capacity (1 024 SIMDs, one wave-instruction per 4 cycles) its
    kernels use — SQ_INSTS_VALU and the clock from the committed counter passes (profiles/valu.json), the time from this run.
    SURVEY.md 8d names FP64 VALU as the close second bound; for every kernel but bsc_call_kernel it is the one that binds.
  * cpu_baseline (rank 0, N = 1): the CPU oracle's libm flavour (= the reference's arithmetic; "port") on this host's
    cores over a bounded sample of the same contig: SURVEY.md 8d's three timings, medians of repetitions of >= 1 s.

--config 3 = BASELINE.json configs[2]: a human-scale genome (24 contigs, 3.1 G positions, 30x, 1 % N-runs), contigs
  assigned to ranks by longest-processing-time (the reference is run one process per contig: README.md:73-76), each
  contig HBM-resident and walked in 4 Mi-position windows through the fused chain with statistics; at the end the ranks
  all-reduce the counter block and the statistics block (RCCL).  A step = one pass over the rank's contigs; value =
  all positions / max-over-ranks time (strong scaling).  --rank-of R [--rank-index I] runs the share rank I of R would
  own on THIS process alone (what one GPU of an R-GPU run does).
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_COVERED = 305
ALGO_BYTES_UNCOVERED = 105
CHAIN_BYTES = 105 + 64  # fused chain: pile-up + reference code in, bsc_vcf_core out
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
VALU_SIMDS = 1024  # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_INST = 4  # a wave64 instruction occupies its SIMD's VALU for 4 cycles, FP64 included (78.6 TFLOP/s vector FP64)
SEED = 88172645463325252
KERNEL_SOURCES = ("kernels.hip", "callmath.h", "call_body.inc", "call_summary.inc", "bsmath.h", "bsmath_tables.h", "devtables.h")
READS_SOURCES = ("fused.hip", "accdev.h", "accumulate.hip", "callmath.h", "call_body.inc", "call_summary.inc", "sitestats_dev.h", "bsmath.h",
                 "bsmath_tables.h", "devtables.h")
CHAIN_SOURCES = ("fused.hip", "accdev.h", "callmath.h", "call_body.inc", "call_summary.inc", "sitestats_dev.h", "bsmath.h", "bsmath_tables.h",
                 "devtables.h")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3), help="2: configs[1], 50 Mb contig (default); 3: configs[2], human-scale genome")
    ap.add_argument("--sites", type=int, default=50_000_000, help="positions per rank (config 2: 50 Mb)")
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--cpu-sites", type=int, default=32_000_000, help="sample size of the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chain", action="store_true", help="skip the fused-chain measurement (roofline_chain)")
    ap.add_argument("--no-reads", action="store_true", help="skip the reads-in measurements (roofline_accumulate, roofline_reads)")
    ap.add_argument("--rank-of", type=int, default=0, help="config 3: run the share of one rank of this many, on this process alone")
    ap.add_argument("--rank-index", type=int, default=0)
    ap.add_argument("--window", type=int, default=0, help="config 3: positions per window (0 = genome.window_for: the largest whole number of resident-wave rounds within 4 Mi)")
    ap.add_argument("--mem-gb", type=float, default=0.0, help="config 3: HBM budget for resident contigs (0 = 80 %% of free)")
    ap.add_argument("--dbsnp", action="store_true", help="config 3 -> BASELINE.json configs[4]: a synthetic dbSNP index (1 site / 300 bp, "
                    "10 %% fq_mask) is written, read back through the library's reader and its flags drive the chain")
    ap.add_argument("--genome-scale", type=float, default=1.0, help="config 3: scale every contig length (rehearsals)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process never touches a GPU; it starts one rank per GPU as a fresh
        # child (torch.distributed.run), relays the child's output (rank 0's JSON line) and exits with its code.
        sys.exit(launch_ranks(args.gpus))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_REHEARSAL=1: all ranks share cuda:0 and talk over gloo — lets the N > 1 code path be rehearsed on a
        # one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver; the numbers mean nothing.
        rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
        if rehearsal:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    env = {"world": world, "rank": rank, "dist": dist, "dev": torch.device("cuda", torch.cuda.current_device())}
    res = run_config3(args, env) if args.config == 3 else run_config2(args, env)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(n):
    """One process per GPU (the reference's unit of parallelism is one process per contig, README.md:73-76): the same
    command line under torch.distributed.run on 127.0.0.1 and a free port.  Called before anything initialises the GPU."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def barrier(env):
    import torch

    if env["dist"] is not None:
        env["dist"].barrier()
    torch.cuda.synchronize()


def max_over_ranks(env, dt):
    """The job's time = the slowest rank's.  Also leaves every rank's own time in env["rank_s"] (one all-reduce of a world-sized
    vector, each rank filling its own slot) for the line's max / mean — a strong-scaling run is as fast as its largest share."""
    import torch

    if env["dist"] is None:
        env["rank_s"] = [dt]
        return dt
    t = torch.zeros(env["world"], dtype=torch.float64, device=env["dev"])
    t[env["rank"]] = dt
    env["dist"].all_reduce(t)
    env["rank_s"] = [float(v) for v in t.cpu()]
    return max(env["rank_s"])


def rank_time_summary(env):
    rs = env.get("rank_s") or []
    if not rs:
        return None
    mean = sum(rs) / len(rs)
    return {"max_s": max(rs), "mean_s": mean, "max_over_mean": max(rs) / mean if mean > 0 else None, "per_rank_s": [round(v, 6) for v in rs]}


# ---------------------------------------------------------------------------------------------------------------------
def run_config2(args, env):
    import numpy as np
    import torch

    import bs_call_amd as B

    world, rank, dist, dev = env["world"], env["rank"], env["dist"], env["dev"]
    n = args.sites
    caller = B.SiteCaller(device=dev.index)
    d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)  # two records behind the contig: chain context
    d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
    d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # rank r calls contig r of the synthetic genome: same generator, disjoint site range
    first_site = rank * (n + 64)
    caller.synth_device(SEED + 2, first_site, n + 2, args.coverage, d_cts.data_ptr(), d_ref.data_ptr(), 0, stream)
    torch.cuda.synchronize()

    def step():
        caller.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, stream)

    caller.set_profiling(True)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    caller.reset_stats()

    kernel_ms, fisher_ms = [], []
    barrier(env)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # queued back to back: nothing in the loop waits for the device
    stats = torch.from_numpy(caller.stats_vector()).to(dev)  # syncs the stream
    if dist is not None:
        dist.all_reduce(stats)  # the only collective: per-rank counters (RCCL)
    barrier(env)
    dt = max_over_ranks(env, time.perf_counter() - t0)
    # device times of the timed launches, from the HIP events the library recorded on the launch stream (the last 32 are kept)
    hist = caller.kernel_ms_history(args.steps)
    kernel_ms, fisher_ms = [h[0] for h in hist], [h[1] for h in hist]

    stats = stats.cpu().numpy()
    total_sites = int(stats[0])
    covered = int(stats[1])
    assert total_sites == n * world * args.steps, (total_sites, n, world, args.steps)
    value = total_sites / dt
    res = None
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        cov_frac = covered / total_sites
        algo_bytes = n * (cov_frac * ALGO_BYTES_COVERED + (1.0 - cov_frac) * ALGO_BYTES_UNCOVERED)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        res = {
            "metric": "genome positions called/sec",
            "value": value,
            "unit": "positions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "gbases_per_hour": value * 3.6e-6,
            "rank_time": rank_time_summary(env),  # max / mean over the ranks of the timed region (the value is positions / max)
            "config": {
                "workload": "configs[1]: synthetic chr22-sized contig, %d positions at %dx WGBS (L-pileup generator), "
                "pile-ups resident in HBM, one contig per GPU" % (n, args.coverage),
                "positions_per_gpu": n,
                "coverage": args.coverage,
                "sharding": "contig per rank, no data-path collective; one RCCL all-reduce of 13 counters at the end",
                "covered_fraction": cov_frac,
                "het_call_fraction": float(stats[12]) / max(covered, 1),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "bsc_call_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": profiled_traffic(n, args.coverage),
                "valu": valu_block(n, args.coverage, "call", k_ms + float(np.mean(fisher_ms))),
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms_avg": k_ms,
                "kernel_ms_min": float(np.min(kernel_ms)),
                "fisher_kernel_ms_avg": float(np.mean(fisher_ms)),
                "positions_per_s_kernel_only": n / (k_ms * 1e-3),
            },
        }
    if world == 1 and not args.no_chain:
        res["roofline_chain"] = chain_roofline(args, caller, d_cts, d_ref, n, first_site)
    if world == 1 and not args.no_reads:
        res.update(reads_rooflines(args, caller))
    reads_sample = res.pop("_reads_sample", None) if res else None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res.update(cpu_baseline(args, d_cts, d_ref, d_out, d_skip))
        if reads_sample is not None:
            check_reads_sample(args, res, reads_sample)
    if rank == 0:
        # Last (it overwrites d_out): the same bytes moved by a kernel that does nothing else — what the memory system
        # delivers for this 1 : 2 read : write mix (HBM3E writes stream slower than reads), measured live on this GPU.
        probe_ms = caller.stream_probe_ms(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 5, stream)
        if probe_ms > 0.0:
            r = res["roofline"]
            r["stream_probe"] = {
                "what": "copy kernel with the calling kernel's traffic and tile shape, no arithmetic (csrc/probe.hip); average of 5 launches queued back to back behind 3 untimed ones",
                "ms": probe_ms,
                "GBps": r["algorithmic_bytes_per_launch"] / (probe_ms * 1e-3) / 1e9,
                "kernel_frac_of_probe": probe_ms / r["kernel_ms_avg"],
            }
    caller.close()
    return res


WARM_MS = 40.0  # device time of untimed launches in front of the legs below: after an idle stretch (host-side set-up) the first
# ~20 ms of launches run below the steady clock (tools/warm_hist.py) — it matters for launches shorter than a few ms


def timed_launches(launch, last_ms, reps):
    """`launch()` queues one launch and waits for it; `last_ms()` is its device time (HIP events inside the library).  Untimed
    launches until WARM_MS of device time have run (2 .. 200 of them), then `reps` timed ones: their device times."""
    warm, k = 0.0, 0
    while k < 2 or (warm < WARM_MS and k < 200):
        launch()
        warm += last_ms()
        k += 1
    ms = []
    for _ in range(reps):
        launch()
        ms.append(last_ms())
    return ms


def chain_roofline(args, caller, d_cts, d_ref, n, first_site):
    """The fused print-side chain over the same resident contig (one block, one call), with statistics: device time from
    HIP events on the launch stream (bsc_last_chain_ms), 105 + 64 algorithmic bytes per position."""
    import numpy as np
    import torch

    dev = d_cts.device
    d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    wall = []

    def launch():
        t0 = time.perf_counter()
        caller.chain_device(d_cts.data_ptr(), d_ref.data_ptr(), 1, n, 0, n, d_core.data_ptr(), with_stats=True, stream=stream)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)

    reps = min(args.steps, 10)
    ms = timed_launches(launch, caller.last_chain_ms, reps)
    k_ms = float(np.mean(ms))
    wall_s = float(np.median(wall[-reps:]))
    achieved = n * CHAIN_BYTES / (k_ms * 1e-3) / 1e9
    records = int(d_core.view(n, 64)[:, 4].sum())
    return {
        "bound": "valu_issue",
        "kernel": "bsc_chain_kernel_t (bsc_chain_device)",
        "what": "pile-up -> call -> VCF record -> site statistics in one pass; gt_meth never reaches HBM (the unfused chain "
        "moves 630 B per position)",
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "traffic": profiled_traffic(n, args.coverage, "chain"),
        "valu": valu_block(n, args.coverage, "chain", k_ms),
        "algorithmic_bytes_per_launch": n * CHAIN_BYTES,
        "kernel_ms_avg": k_ms,
        "positions_per_s": n / wall_s,  # per call as the host sees it (launch + one wait), median: round 2's definition of this key
        "positions_per_s_device": n / (k_ms * 1e-3),  # HIP events around the launches (round 3 reported this one under the key above)
        "records_written_fraction": records / n,
        "note": "bound by VALU instruction issue (FP64 model + record formation), not by HBM: `frac` (of 8 TB/s) is low by "
        "construction, `valu.frac` is the fraction of the bound that binds",
    }


def reads_rooflines(args, caller):
    """HOT LOOP A alone (bsc_accumulate_device) and the reads-in chain (bsc_reads_chain_device, with statistics) over one block
    of device-resident L-reads the size of the benchmarked contig; a sample of both outputs checked against the CPU oracle."""
    import numpy as np
    import torch

    import bs_call_amd as B
    from bs_call_amd import reads as R

    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    x, chunk = 1000, 1_000_000
    tpl, seq, y = R.synth_block(SEED + 2, x, args.sites, args.coverage, chunk=chunk)
    n = y - x + 1
    n_pad = (n + 63) // 64 * 64
    ref = B.synth_ref_host(SEED + 2, x, n + 2)
    d_tpl = torch.from_numpy(tpl.view(np.uint8).reshape(-1)).to(dev)
    d_seq = torch.from_numpy(seq).to(dev)
    d_ref = torch.from_numpy(ref).to(dev)
    d_pile = torch.empty(n_pad * 104, dtype=torch.uint8, device=dev)
    d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    def launch_acc():
        caller.accumulate_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_pile.data_ptr(), stream)
        caller.block_status(stream)

    def launch_rc():
        caller.reads_chain_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_ref.data_ptr(), d_core.data_ptr(),
                                  with_stats=True, stream=stream)
        caller.block_status(stream)

    reps = min(args.steps, 10)
    acc_ms = timed_launches(launch_acc, caller.last_accumulate_ms, reps)
    rc_ms = timed_launches(launch_rc, caller.last_reads_chain_ms, reps)
    # a sample for the checker (cpu_baseline): positions well inside the first chunk are covered by that chunk's templates only
    m = min(chunk, args.sites)
    keep = (m - 400) if args.sites > m else n
    sample = {"x": x, "m": m, "chunk": chunk, "keep": keep, "pile": d_pile[: keep * 104].cpu().numpy(), "core": d_core[: (keep - 8) * 64].cpu().numpy()}
    bcf_leg = bcf_roofline(caller, d_tpl, len(tpl), d_seq, seq.size, x, y, d_ref, d_core, n, reps, stream)
    del d_pile
    raw_legs = raw_rooflines(args, caller, tpl, seq, d_seq, x, y, ref, reps)
    bytes_in = R.algorithmic_bytes_in(tpl, seq)
    a_ms, r_ms = float(np.mean(acc_ms)), float(np.mean(rc_ms))
    a_bytes, r_bytes = bytes_in + n * 104, bytes_in + n + n * 64
    block = "one block of %d positions at %dx: %d templates, %d bases, resident in HBM" % (n, args.coverage, len(tpl), seq.size)
    return {
        "roofline_accumulate": {
            "bound": "valu_issue",
            "kernel": "bsc_bin_count_kernel + rocPRIM prefix sum + bsc_bin_scatter_kernel + bsc_accumulate_kernel (bsc_accumulate_device)",
            "what": "HOT LOOP A, reads -> pile-up (reference src/call_genotypes.c:180-226, serial on its process thread); " + block,
            "achieved": a_bytes / (a_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": a_bytes / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "traffic": profiled_traffic(args.sites, args.coverage, "accumulate"),
            "valu": valu_block(args.sites, args.coverage, "accumulate", a_ms),
            "algorithmic_bytes_per_launch": a_bytes,
            "algorithmic_bytes_per_position": a_bytes / n,
            "stage_ms_avg": a_ms,
            "stage_ms_min": float(np.min(acc_ms)),
            "positions_per_s": n / (a_ms * 1e-3),
            "bases_per_s": seq.size / (a_ms * 1e-3),
            "first_chunk_equals_oracle": None,
            "note": "latency and instruction issue (one byte load per read and 64-position tile) share this stage with the HBM writes "
            "of the 104-byte pile-ups; neither bound is reached: both fractions are reported.  The generator hands over whether read 0 "
            "was walked (bsc_template.flags, as bsc_prepare_templates and the glue do); without the flag the grouping fetches one byte of "
            "every read 0 itself (+ 0.18 ms)",
        },
        "roofline_reads": {
            "bound": "valu_issue",
            "kernel": "bsc_bin_count_kernel + prefix sum + bsc_bin_scatter_kernel + bsc_accumulate_kernel_t<summary> + bsc_chain_kernel_t<.., summary-in> "
            "(bsc_reads_chain_device, its default two-kernel form), with statistics",
            "what": "reads -> site summaries (16-bit class counts + the per-site summary of src/call_genotypes.c:44-59, 48 B per position through HBM) -> call -> "
            "VCF record -> site statistics; gt_meth never leaves the registers; " + block,
            "achieved": r_bytes / (r_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": r_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "traffic": profiled_traffic(args.sites, args.coverage, "reads"),
            "valu": valu_block(args.sites, args.coverage, "reads", r_ms),
            "algorithmic_bytes_per_launch": r_bytes,
            "algorithmic_bytes_per_position": r_bytes / n,
            "stage_ms_avg": r_ms,
            "stage_ms_min": float(np.min(rc_ms)),
            "positions_per_s": n / (r_ms * 1e-3),
            "records_written_fraction": int(d_core.view(n, 64)[:, 4].sum()) / n,
            "first_chunk_records_equal_oracle": None,
            "note": "bound by VALU instruction issue (FP64 model + record formation + the pile-up walk), not by HBM: `valu.frac` is the "
            "fraction of the bound that binds.  traffic / algorithmic bytes: the 48-byte summaries are written by one kernel and read by "
            "the next (round 4: 88 bytes, 2.96 x)",
        },
        "roofline_bcf": bcf_leg,
        "roofline_prep": raw_legs["roofline_prep"],
        "raw_to_bcf": raw_legs["raw_to_bcf"],
        "_reads_sample": sample,
    }


def _committed(name, key, sources):
    """a per-launch figure of profiles/<name>.json[key], valid while the kernel sources it was measured on are unchanged"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            v = json.load(f).get(key)
        if not v or v.get("kernel_source_sha256_16") != kernel_source_hash(sources):
            return None
        return v
    except (OSError, ValueError, KeyError):
        return None


def raw_rooflines(args, caller, tpl, seq, d_seq, x, y, ref, reps):
    """Round 5's device stages in the driver-run line (VERDICT r05, row d).
    roofline_prep: device-resident RAW L-reads of the benchmarked contig (the span is the length; every 50th read 0 carries a 2-base deletion
    and a 1-base insertion, so that the list logic and the padded copy run) -> prepared reads (bsc_prepare_templates_device, csrc/prepdev.hip:
    process_template_vector, reference src/process_template.c:36-111), without and with the read profile (meth_profile, src/meth_profile.c:48-77
    — the form the pipeline runs).  raw_to_bcf: the same block from raw templates resident in HBM to its BCF stream resident in HBM
    (bsc_block_bcf_rawdev_keep: pre-processing + read profile + grouping + walk + chain + encoder between two events on the library's stream)."""
    import ctypes as C

    import numpy as np
    import torch

    import bs_call_amd as B
    from bs_call_amd import _lib, vcf
    from bs_call_amd.abi import MISMS, PREP_PARAMS, PREP_STATS, RAW_TEMPLATE, TEMPLATE
    from bs_call_amd.caller import ReadProfile, _ptr

    dev = d_seq.device
    stream = torch.cuda.current_stream().cuda_stream
    raw = np.zeros(len(tpl), dtype=RAW_TEMPLATE)
    for f in ("pos", "len", "off", "mapq", "orientation", "bs_strand"):
        raw[f] = tpl[f]
    raw["reference_span"] = tpl["len"]
    sel = np.nonzero((np.arange(len(raw)) % 50 == 0) & (raw["len"][:, 0] >= 60))[0]
    ms = np.zeros(2 * len(sel), dtype=MISMS)
    ms["type"][0::2], ms["position"][0::2], ms["size"][0::2] = 1, 20, 2
    ms["type"][1::2], ms["position"][1::2], ms["size"][1::2] = 2, 40, 1
    raw["n_misms"][sel, 0] = 2
    raw["misms_off"][sel, 0] = 2 * np.arange(len(sel))
    raw["reference_span"][sel, 0] += 1
    y_raw = int(y) + 1  # a read with the deletion reaches one position further
    par = np.zeros(1, dtype=PREP_PARAMS)
    par["min_qual"] = args.min_qual if hasattr(args, "min_qual") else 20
    ins_pad = 2 * len(sel)
    cap = int(seq.size) + ins_pad + 16
    up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
    d_raw, d_ms = up(raw), up(ms)
    d_tpl = torch.empty(len(raw) * TEMPLATE.itemsize, dtype=torch.uint8, device=dev)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    ref_raw = B.synth_ref_host(SEED + 2, x, y_raw - x + 3)
    d_ref = up(ref_raw)
    used, st = C.c_uint64(0), np.zeros(1, dtype=PREP_STATS)
    L = caller._L
    alg = int(seq.size) + len(raw) * 72 + len(ms) * 12 + len(raw) * 40  # + the prepared bytes, added below
    legs = {}
    for with_profile in (False, True):
        prof = ReadProfile(cap=4096)
        pf = _lib.ReadProfile(d_ref.data_ptr(), x, y_raw - x + 3, prof.counts.ctypes.data, prof.counts.shape[0], 0) if with_profile else None

        def call():
            if pf is not None:
                pf.used = 0
            rc = L.bsc_prepare_templates_device(caller._h, d_raw.data_ptr(), len(raw), d_seq.data_ptr(), seq.size, d_ms.data_ptr(), len(ms), _ptr(par),
                                                d_tpl.data_ptr(), d_out.data_ptr(), cap, C.byref(used), _ptr(st), None if pf is None else C.byref(pf), stream)
            assert rc == 0, L.bsc_last_error()

        for _ in range(2):
            call()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
        for k in range(reps):
            ev[2 * k].record()
            call()  # (it waits for the prepared size itself)
            ev[2 * k + 1].record()
        torch.cuda.synchronize()
        msv = [ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(reps)]
        k_ms = float(np.mean(msv))
        a = alg + int(used.value)
        key = "with_read_profile" if with_profile else "plain"
        prof_v = _committed("traffic.json", "prep_profile" if with_profile else "prep", ("prepdev.hip",))
        valu_v = _committed("valu.json", "prep_profile" if with_profile else "prep", ("prepdev.hip",))
        legs[key] = {
            "stage_ms_avg": k_ms, "stage_ms_min": float(np.min(msv)), "algorithmic_bytes_per_launch": a, "achieved": a / (k_ms * 1e-3) / 1e9,
            "frac": a / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "bases_per_s": seq.size / (k_ms * 1e-3), "positions_per_s": args.sites / (k_ms * 1e-3),
            "traffic": None if prof_v is None else int(prof_v["hbm_bytes_per_launch"] * (args.sites / prof_v["sites"])),
            "valu_instructions_per_launch": None if valu_v is None else int(valu_v["valu_insts_per_launch"] * (args.sites / valu_v["sites"])),
        }
        if not with_profile:  # the first templates against the host form (csrc/prep.c)
            from bs_call_amd.caller import prepare_templates

            m = min(100_000, len(raw))
            h_tpl, h_seq, _ = prepare_templates(raw[:m], seq, ms)
            legs[key]["first_templates_equal_host_form"] = bool(d_tpl[: m * 40].cpu().numpy().tobytes() == h_tpl.tobytes()
                                                                 and d_out[: int(h_seq.size)].cpu().numpy().tobytes() == h_seq.tobytes())
    head = legs["with_read_profile"]
    prep_leg = {
        "bound": "valu_issue",
        "kernel": "bsc_prep_plan_kernel + rocPRIM prefix sum + bsc_prep_copy_kernel + the read profile's pass (max-scan, bsc_prep_refmask_kernel, bsc_prep_profile_kernel) (bsc_prepare_templates_device)",
        "what": "raw templates + reads + mismatch lists resident in HBM -> prepared templates + reads (trims, soft clips, mate overlap, indel normalisation; the "
        "base counters; with the read profile as the pipeline runs it): %d templates, %d bases, %d list entries" % (len(raw), seq.size, len(ms)),
        "achieved": head["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": head["frac"], "traffic": head["traffic"],
        "stage_ms_avg": head["stage_ms_avg"], "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"],
        "algorithmic_bytes_per_position": head["algorithmic_bytes_per_launch"] / args.sites,
        "with_read_profile": legs["with_read_profile"], "plain": legs["plain"],
        "note": "torch events of the stream the launches are queued on; each call waits once for the prepared size (one host round trip inside the "
        "timed region).  The headline is the form with the read profile: what pipeline.run / bam2bcf run",
    }
    # ---- raw templates resident -> BCF stream resident ----
    caller.set_profiling(True)
    prof = ReadProfile(cap=4096)
    ids = _lib.BcfIds()
    L.bsc_bcf_default_ids(C.byref(ids))
    n = y_raw - x + 1
    dev_cap = n * 96 + 4096
    vp = _lib.VcfParams(0, 1, 0xFFFFFFFF)
    ref_pin = L.bsc_alloc_host(len(ref_raw))  # page-locked: the block's reference codes cross PCIe inside the timed region (1 B per position)
    C.memmove(ref_pin, ref_raw.ctypes.data, len(ref_raw))
    nb, nr_ = C.c_uint64(0), C.c_uint64(0)
    st2 = np.zeros(1, dtype=PREP_STATS)
    msv, wall = [], []
    for it in range(2 + reps):
        pf = _lib.ReadProfile(None, 0, 0, prof.counts.ctypes.data, prof.counts.shape[0], 0)
        t0 = time.perf_counter()
        rc = L.bsc_block_bcf_rawdev_keep(caller._h, d_raw.data_ptr(), len(raw), d_seq.data_ptr(), seq.size, d_ms.data_ptr(), len(ms), ins_pad, _ptr(par), x, y_raw,
                                         ref_pin, None, C.byref(vp), 1 if it == 0 else 0, 0, C.byref(ids), None, dev_cap, C.byref(nb), C.byref(nr_), _ptr(st2),
                                         C.byref(pf))
        assert rc >= 0, L.bsc_last_error()
        dt = time.perf_counter() - t0
        f = C.c_float(0)
        assert L.bsc_last_raw_block_ms(caller._h, C.byref(f)) == 0, L.bsc_last_error()
        if it >= 2:
            msv.append(f.value)
            wall.append(dt)
    caller.set_profiling(False)
    # the first records: the stream's head against the host encoder over the packed records of the same block
    head_bytes = min(int(nb.value), 8 << 20)
    got = np.empty(head_bytes, dtype=np.uint8)
    assert L.bsc_bcf_stream_read(caller._h, 0, head_bytes, got.ctypes.data) == 0 and L.bsc_synchronize(caller._h) == 0
    # ... of a sub-block made of the first templates: its records left of where the next template starts are the big block's
    K = min(40_000, len(raw))
    p_safe = (int(raw["pos"][K, 0] or raw["pos"][K, 1]) - 10) if K < len(raw) else 1 << 62
    sub = raw[:K]
    seq_end = int((sub["off"].astype(np.int64) + sub["len"]).max())
    n_ms_sub = 2 * int((sel < K).sum())
    y_sub = int((sub["pos"].astype(np.int64) + sub["reference_span"]).max())
    recs, _st = caller.block_records_raw(sub, seq[:seq_end], ms[:n_ms_sub], x, y_sub, ref_raw[: y_sub - x + 3], min_qual=int(par["min_qual"][0]), with_stats=False)
    recs = recs[recs["core"]["pos"] <= p_safe]
    want = vcf.bcf_block(recs, 0)
    L.bsc_free_host(ref_pin)
    k_ms = float(np.mean(msv))
    raw_leg = {
        "what": "one block, raw templates + reads + lists resident in HBM -> its BCF stream resident in HBM: pre-processing with the read profile, grouping, "
        "walk, chain with statistics' kernels, the encoder over the chain's per-position arrays (bsc_block_bcf_rawdev_keep): what the device reader's blocks go "
        "through; %d positions, %d templates" % (n, len(raw)),
        "device_ms_avg": k_ms, "device_ms_min": float(np.min(msv)), "call_wall_ms_median": float(np.median(wall)) * 1e3,
        "positions_per_s": n / (k_ms * 1e-3), "positions_per_s_call_wall": n / float(np.median(wall)),
        "bcf_bytes": int(nb.value), "records": int(nr_.value), "bytes_out_per_position": int(nb.value) / n,
        "first_records_equal_host_encoder": bool(len(want) > 0 and got[: min(len(want), head_bytes)].tobytes() == want[: min(len(want), head_bytes)]),
        "first_bytes_compared": int(min(len(want), head_bytes)),
        "note": "HIP events on the library's own stream, first pre-processing launch to the encoder's last (bsc_last_raw_block_ms); inside: ONE host wait "
        "(the block's end; round 5 waited for the prepared size as well), the upload of the block's reference codes (1 B per position, page-locked) and the read profile's counts coming back",
    }
    return {"roofline_prep": prep_leg, "raw_to_bcf": raw_leg}


def bcf_traffic(n_rec):
    """HBM bytes per launch of the encoder from the committed PMC passes (profiles/traffic.json["bcf"], tools/pmc_bcf.sh): measured per
    record at configs[1] size, scaled to this block's record count; None once csrc/bcfdev.hip has changed."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            v = json.load(f).get("bcf")
        if not v or v.get("kernel_source_sha256_16") != kernel_source_hash(("bcfdev.hip",)):
            return None
        return int(v["hbm_bytes_per_record"] * n_rec)
    except (OSError, ValueError, KeyError):
        return None


def bcf_roofline(caller, d_tpl, nr, d_seq, seq_bytes, x, y, d_ref, d_core, n, reps, stream):
    """The tail of record formation on the device (csrc/bcfdev.hip): the block's packed written records -> the BCF stream bcf_write would
    emit for them (reference src/print_vcf.c:160-222,267-378), timed with events on the launch stream; the first records compared with
    the host encoder's bytes (csrc/bcf.c, pinned record by record to the independent encoder of oracle/py_bcf.py by tests/test_bcf.py)."""
    import numpy as np
    import torch

    import bs_call_amd as B
    from bs_call_amd import vcf

    dev = d_core.device
    d_aux = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    caller.reads_chain_device(d_tpl.data_ptr(), nr, d_seq.data_ptr(), seq_bytes, x, y, d_ref.data_ptr(), d_core.data_ptr(), d_aux=d_aux.data_ptr(),
                              with_stats=False, stream=stream)
    caller.block_status(stream)
    cap_rec = int(n * 0.75) + 4096
    d_rec = torch.empty(cap_rec * 128, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    caller.vcf_compact_device(d_core.data_ptr(), d_aux.data_ptr(), 0, n, d_rec.data_ptr(), cap_rec, d_cnt.data_ptr(), stream=stream)
    n_rec = int(d_cnt.item())
    if n_rec > cap_rec:
        return {"skipped": "more written records (%d) than the leg's buffer holds (%d)" % (n_rec, cap_rec)}
    cap = n_rec * 136 + 4096
    d_bcf = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_tot = torch.zeros(3, dtype=torch.int64, device=dev)

    def launch():
        caller.bcf_block_device(d_rec.data_ptr(), d_cnt.data_ptr(), cap_rec, 0, d_bcf.data_ptr(), cap, d_tot.data_ptr(), stream=stream)

    for _ in range(4):
        launch()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        launch()
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]
    nbytes, refused = int(d_tot[0].item()), int(d_tot[1].item())

    # the form bsc_block_bcf runs: straight from the chain's per-position arrays, no packing pass — against packing + encoding
    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        e[0].record()
        for k in range(reps):
            fn()
            e[k + 1].record()
        torch.cuda.synchronize()
        return float(np.mean([e[k].elapsed_time(e[k + 1]) for k in range(reps)]))

    d_bcf2 = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_tot2 = torch.zeros(3, dtype=torch.int64, device=dev)
    # round 6: the chain leaves the records' BCF2 lengths in a byte per position; the size pass reads those (what the block entries run)
    d_len = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    caller.reads_chain_len_device(d_tpl.data_ptr(), nr, d_seq.data_ptr(), seq_bytes, x, y, d_ref.data_ptr(), d_core.data_ptr(), d_aux.data_ptr(),
                                  d_len.data_ptr(), with_stats=False, stream=stream)
    caller.block_status(stream)
    sites_rec_ms = timed(lambda: caller.bcf_sites_device(d_core.data_ptr(), d_aux.data_ptr(), n, 0, d_bcf2.data_ptr(), cap, d_tot2.data_ptr(), stream=stream))
    sites_ms = timed(lambda: caller.bcf_sites_len_device(d_core.data_ptr(), d_aux.data_ptr(), d_len.data_ptr(), n, 0, d_bcf2.data_ptr(), cap, d_tot2.data_ptr(),
                                                         stream=stream))
    pack_ms = timed(lambda: caller.vcf_compact_device(d_core.data_ptr(), d_aux.data_ptr(), 0, n, d_rec.data_ptr(), cap_rec, d_cnt.data_ptr(), stream=stream))
    sites_same = int(d_tot2[0].item()) == nbytes and int(d_tot2[2].item()) == n_rec and bool(torch.equal(d_bcf[:nbytes], d_bcf2[:nbytes]))
    sites_alg = n + (n + 128 * n_rec) + nbytes  # size pass: the chain's byte per position; write kernel: that byte again, 128 B of a written record, the stream
    del d_aux, d_bcf2
    m = min(200_000, n_rec)
    want = vcf.bcf_block(d_rec[: m * 128].cpu().numpy().view(B.VCF_REC), 0)
    same = nbytes <= cap and refused == 0 and d_bcf[: len(want)].cpu().numpy().tobytes() == want
    k_ms = float(np.mean(ms))
    alg = n_rec * 128 * 2 + nbytes  # both kernels read the records, the stream is written once
    sites_v = _committed("traffic.json", "bcf_sites", ("bcfdev.hip",))
    return {
        "bound": "hbm",
        "kernel": "bsc_bcf_size_bytes_kernel + rocPRIM prefix sum + bsc_bcf_write_kernel_t over the chain's per-position arrays, gated and sized by the "
        "chain's byte per position (bsc_bcf_sites_len_device: the form bsc_block_bcf[_raw[dev]] runs; no packing pass)",
        "what": "the reads-in chain's per-position arrays (64-byte record + 64-byte aux per position) -> BCF2 records (typed values of "
        "src/print_vcf.c:160-222,267-378 behind bcf_write's fixed fields): %d positions, %d written records, resident in HBM" % (n, n_rec),
        "achieved": sites_alg / (sites_ms * 1e-3) / 1e9,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": sites_alg / (sites_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "traffic": None if sites_v is None else int(sites_v["hbm_bytes_per_position"] * n),
        "algorithmic_bytes_per_launch": sites_alg,
        "bcf_bytes": nbytes,
        "bytes_per_record": nbytes / max(n_rec, 1),
        "stage_ms_avg": sites_ms,
        "records_per_s": n_rec / (sites_ms * 1e-3),
        "positions_per_s": n / (sites_ms * 1e-3),
        "first_records_equal_host_encoder": bool(same and sites_same),
        "packed_form": {
            "kernel": "the same two kernels over packed written records (bsc_bcf_block_device, behind bsc_vcf_compact_device): round 5's headline",
            "stage_ms_avg": k_ms,
            "stage_ms_min": float(np.min(ms)),
            "algorithmic_bytes_per_launch": alg,
            "achieved": alg / (k_ms * 1e-3) / 1e9,
            "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "traffic": bcf_traffic(n_rec),
            "records_per_s": n_rec / (k_ms * 1e-3),
        },
        "sites_form": {
            "kernel": "the same two kernels over the chain's per-position arrays (bsc_bcf_sites_device: what bsc_block_bcf runs), no packing pass",
            "stage_ms_avg": sites_ms,
            "algorithmic_bytes_per_launch": sites_alg,
            "achieved": sites_alg / (sites_ms * 1e-3) / 1e9,
            "frac": sites_alg / (sites_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "size_pass_over_the_records_ms": sites_rec_ms,
            "sector_bytes_per_launch": sites_alg,
            "sector_frac": sites_alg / (sites_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "sector_note": "with the chain's byte per position a position without a record is not touched (before: the 64-byte sector around its first 16 "
            "bytes), and a written record is two whole sectors: what the memory system moves is the algorithmic bytes, `sector_frac` = `frac`",
            "packing_pass_ms_it_replaces": pack_ms,
            "packing_plus_encoding_ms": pack_ms + k_ms,
            "same_stream": bool(sites_same),
        },
        "note": "events of torch's current stream, which is the stream the launches are queued on; the written records are read once (the size pass reads "
        "a byte per position); packed_form reads them twice (sizes, then bytes)",
    }


def _code_only(text):
    """C / HIP source without comments and with white space collapsed: what the compiler sees, as far as a traffic figure cares."""
    import re

    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return " ".join(text.split())


def kernel_source_hash(sources=KERNEL_SOURCES):
    """Tag of the kernel sources a PMC traffic figure belongs to: comments and white space do not enter it."""
    h = hashlib.sha256()
    for f in sources:
        with open(os.path.join(ROOT, "bs_call_amd", "csrc", f), "r", encoding="utf-8") as fh:
            h.update(_code_only(fh.read()).encode("utf-8"))
    return h.hexdigest()[:16]


def profiled_traffic(n, coverage, which=None):
    """HBM bytes per bsc_call_kernel launch from the committed PMC passes (profiles/traffic.json, written by
    tools/make_traffic_json.py from `tools/profile_bench.sh`: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command, FETCH_SIZE doubled per the gfx950 rule).  Counters cannot be read inside this process; None when
    the workload differs or the kernel sources have changed since the passes were taken."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        want = kernel_source_hash()
        if which == "chain":  # the fused chain kernel's own passes (tools/bench_chain.py under --pmc), own source hash
            t, want = t["chain"], kernel_source_hash(CHAIN_SOURCES)
        elif which in ("reads", "accumulate"):  # tools/bench_reads.py under --pmc
            t, want = t[which], kernel_source_hash(READS_SOURCES)
        if t.get("positions") == n and t.get("coverage") == coverage and t.get("kernel_source_sha256_16") == want:
            return t["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def valu_block(n, coverage, which, ms):
    """The VALU-issue side of a leg: wave-instructions per launch of all its kernels and the clock its dominant kernel ran at,
    from the committed SQ counter passes (profiles/valu.json, tools/make_valu_json.py — counters cannot be read inside this
    process); frac = instructions x 4 cycles / (1 024 SIMDs x clock x the leg's device time measured HERE).  None when the
    workload differs or the kernel sources have changed since the passes were taken."""
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))[which]
        want = kernel_source_hash({"call": KERNEL_SOURCES, "chain": CHAIN_SOURCES}.get(which, READS_SOURCES))
        if v.get("positions") != n or v.get("coverage") != coverage or v.get("kernel_source_sha256_16") != want or not v.get("clock_ghz"):
            return None
        insts, clk = v["insts_valu_per_launch"], v["clock_ghz"]
        return {"insts_per_launch": insts, "cycles_per_inst": VALU_CYCLES_PER_INST, "simds": VALU_SIMDS, "clock_ghz": clk,
                "frac": insts * VALU_CYCLES_PER_INST / (VALU_SIMDS * clk * 1e9 * ms * 1e-3),
                "what": "SQ_INSTS_VALU of the leg's kernels x 4 cycles / (1 024 SIMDs x clock x this run's device time); clock = "
                "SQ_BUSY_CYCLES / 32 shader engines / duration of the dominant kernel in the counter pass (profiles/valu.json)"}
    except Exception:
        return None


def _median_rate(fn, units, min_s=1.0, reps=5):
    """Median units/s over `reps` repetitions, each repetition looping fn() until >= min_s seconds have passed."""
    rates = []
    for _ in range(reps):
        k, t0 = 0, time.perf_counter()
        while True:
            fn()
            k += 1
            dt = time.perf_counter() - t0
            if dt >= min_s:
                break
        rates.append(units * k / dt)
    return sorted(rates)[len(rates) // 2]


def cpu_baseline(args, d_cts, d_ref, d_out, d_skip):
    """The CPU oracle (libm flavour = the reference's arithmetic) on a bounded sample of the same workload — SURVEY.md
    8d's three timings, single thread and all hardware threads, medians of 5 repetitions of >= 1 s — and, while at it, the
    GPU output of the sample checked against the oracle byte for byte (LIBM flavour when this host's libm is exact)."""
    import numpy as np

    import bs_call_amd as B
    from oracle import loader as O

    m = min(args.cpu_sites, args.sites)
    pile = d_cts[: m * 104].cpu().numpy().view(B.PILEUP)
    ref = d_ref[:m].cpu().numpy()
    tb = O.Tables()
    cores = os.cpu_count() or 1
    L = O.lib()
    out = np.zeros(m, dtype=B.GT_METH)
    skip = np.zeros(m, dtype=np.uint8)
    out[:] = out  # touch the pages so the timings below are compute, not first-touch faults
    call = lambda k, thr: L.orc_call_sites(pile.ctypes.data, ref.ctypes.data, k, tb.ptr, out.ctypes.data, skip.ctypes.data, O.LIBM, thr)
    call(m, -cores)  # warm-up: pages in the library, fills out/skip for (i)
    # (ii) the full per-site path (summary + calc_gt_prob + Fisher + 200-byte record): T threads on contiguous ranges, 1 thread
    m1 = min(m, 4_000_000)
    full_T = _median_rate(lambda: call(m, -cores), m)
    full_1 = _median_rate(lambda: call(m1, 1), m1)
    t_ref = max(1, int(0.4 * (cores - 1)))  # the reference's default split of nproc - 1 extra threads, 4 : 3 : 3 (src/parse_args.c:191-213)
    mi = min(m, 8_000_000)
    full_interleaved = _median_rate(lambda: call(mi, t_ref), mi, reps=3)
    # (i) calc_gt_prob only, on the prepared records of the covered positions
    cov = skip[:m1] == 0
    gt, rf = out[:m1][cov].copy(), ref[:m1][cov].copy()
    model_1 = _median_rate(lambda: L.orc_calc_gt_prob_array(gt.ctypes.data, rf.ctypes.data, len(gt), tb.ptr, O.LIBM), len(gt))
    covT = skip == 0
    gtT, rfT = out[covT].copy(), ref[covT].copy()
    model_T = _median_rate(lambda: L.orc_calc_gt_prob_array_mt(gtT.ctypes.data, rfT.ctypes.data, len(gtT), tb.ptr, O.LIBM, cores), len(gtT))
    # (iii) HOT LOOP A (reads -> pile-up), which the reference runs serially on its process thread
    blk = 400_000
    tpl, seq = B.synth_reads_host(SEED + 2, 1000, blk, args.coverage)
    x, y = 998, int((tpl["pos"] + tpl["len"]).max()) - 1
    sz = y - x + 1
    thr_acc = min(cores, 64)
    pl = np.zeros(sz * thr_acc, dtype=B.PILEUP)
    acc = lambda thr: L.orc_accumulate_mt(tpl.ctypes.data, len(tpl), seq.ctypes.data, x, y, 20, pl.ctypes.data, thr, 2)
    acc_1 = _median_rate(lambda: acc(1), 2 * sz)
    acc_T = _median_rate(lambda: acc(thr_acc), 2 * sz * thr_acc)
    # the reference's own shape: block k+1 is accumulated on ONE thread while 1 + T calc threads walk block k with
    # interleaved striding (src/call_genotypes.c:36-43,180-226,260-272); steady state = the slower of the two
    pile_blk = np.zeros(sz, dtype=B.PILEUP)
    L.orc_accumulate(tpl.ctypes.data, len(tpl), seq.ctypes.data, x, y, 20, pile_blk.ctypes.data)
    ref_blk = B.synth_ref_host(SEED + 2, x, sz)
    out_blk, skip_blk = np.zeros(sz, dtype=B.GT_METH), np.zeros(sz, dtype=np.uint8)

    def ref_shaped():
        th = threading.Thread(target=lambda: acc(1))
        th.start()
        for _ in range(2):
            L.orc_call_sites(pile_blk.ctypes.data, ref_blk.ctypes.data, sz, tb.ptr, out_blk.ctypes.data, skip_blk.ctypes.data, O.LIBM, t_ref)
        th.join()

    ref_rate = _median_rate(ref_shaped, 2 * sz, reps=3)
    # parity of the benchmarked output on a slice of the sample
    exact = O.libm_exact()
    mc = min(m, 1_000_000)
    exp, eskip = O.call_sites(pile[:mc], ref[:mc], tb, O.LIBM if exact else O.BSM, -cores)
    got = d_out[: mc * 200].cpu().numpy().view(B.GT_METH)
    ok = got.tobytes() == exp.tobytes() and (d_skip[:mc].cpu().numpy() == eskip).all()
    return {
        "cpu_baseline": {
            "value": full_T,
            "unit": "positions/s",
            "cores": cores,
            "kind": "port",
            "sample": "first %d positions of the benchmarked contig, oracle libm flavour (the reference's arithmetic); medians of 5 "
            "repetitions of >= 1 s; value = timing (ii), the full per-site path, %d threads on contiguous ranges" % (m, cores),
            "single_thread_value": full_1,
            "timings_positions_per_s": {
                "i_calc_gt_prob_only": {"1_thread": model_1, "%d_threads" % cores: model_T},
                "ii_full_per_site_path": {"1_thread": full_1, "%d_threads_contiguous" % cores: full_T,
                                          "%d_threads_interleaved_as_the_reference" % t_ref: full_interleaved},
                "iii_accumulate_30x": {"1_thread": acc_1, "%d_threads_independent_blocks" % thr_acc: acc_T},
                "reference_shaped": {"value": ref_rate, "what": "serial accumulate of block k+1 beside %d interleaved calc threads on "
                                     "block k (src/call_genotypes.c:180-226,260-272): the serial stage caps it whatever -t says" % t_ref},
            },
            "gpu_output_matches_oracle_on_sample": bool(ok),
            "oracle_flavour_compared": "LIBM" if exact else "BSM",
        },
        "libm_exact": exact,
    }


def check_reads_sample(args, res, sample):
    """Part of the CPU leg: the first chunk of the reads-in measurements' outputs against the oracle chain
    (orc_accumulate -> orc_call_sites -> orc_vcf_block = the reference's process, calc and print threads)."""
    import bs_call_amd as B
    from bs_call_amd import reads as R
    from oracle import loader as O

    x, keep = sample["x"], sample["keep"]
    t1, s1, y1 = R.synth_block(SEED + 2, x, sample["m"], args.coverage, chunk=sample["chunk"])
    _rc, pile = O.accumulate(t1, s1, x, y1, 20)
    ref = B.synth_ref_host(SEED + 2, x, len(pile) + 2)
    exact = O.libm_exact()
    gtm, skip = O.call_sites(pile, ref[: len(pile)], O.Tables(), O.LIBM if exact else O.BSM, -(os.cpu_count() or 1))
    core = O.vcf_block(gtm, skip, ref, x)
    res["roofline_accumulate"]["first_chunk_equals_oracle"] = bool(sample["pile"].tobytes() == pile[:keep].tobytes())
    res["roofline_reads"]["first_chunk_records_equal_oracle"] = bool(sample["core"].tobytes() == core[: keep - 8].tobytes())
    res["roofline_reads"]["oracle_flavour_compared"] = "LIBM" if exact else "BSM"


# ---------------------------------------------------------------------------------------------------------------------
def run_config3(args, env):
    """BASELINE.json configs[2]: 24 human-length contigs at 30x, contig-sharded, fused chain with statistics."""
    import numpy as np
    import torch

    import bs_call_amd as B
    from bs_call_amd import genome, shard

    world, rank, dist, dev = env["world"], env["rank"], env["dist"], env["dev"]
    lengths = [max(1000, int(x * args.genome_scale)) for x in shard.HUMAN_CONTIGS]
    firsts = genome.contig_first_sites(lengths)
    if args.rank_of:
        assert world == 1, "--rank-of emulates one rank of a larger run on a single process"
        v_rank, v_world = args.rank_index, args.rank_of
    else:
        v_rank, v_world = rank, world
    mine = genome.rank_contigs(lengths, v_rank, v_world)
    my_positions = sum(lengths[c] for c in mine)
    free, _total = torch.cuda.mem_get_info()
    budget = int(args.mem_gb * (1 << 30)) if args.mem_gb > 0 else int(free * 0.8)
    groups = genome.batches(mine, lengths, budget)
    caller = B.SiteCaller(device=dev.index)
    caller.set_profiling(os.environ.get("BSC_BENCH_WALK_EVENTS", "0") == "1")  # events between the windows cost a stall each
    if args.window <= 0:
        args.window = genome.window_for(caller)
    stream = torch.cuda.current_stream().cuda_stream
    index = dbsnp_index_for(args, mine, lengths, rank) if args.dbsnp else None
    dt, n_windows, chain_ms, n_db = 0.0, 0, [], 0
    contig_totals = {}

    def flags_of(c):
        nonlocal n_db
        if index is None:
            return None
        n_db += index.load_contig("ctg%d" % c)
        return index.flags(1, lengths[c])

    for gi, group in enumerate(groups):
        resident = [genome.make_resident(caller, c, lengths[c], firsts[c], args.coverage, dev, stream=stream, dbsnp_flags=flags_of(c))
                    for c in group]
        torch.cuda.synchronize()
        if gi == 0:
            for _ in range(args.warmup):
                for rc in resident:
                    genome.walk_contig(caller, rc, args.window, True, stream=stream)
            torch.cuda.synchronize()
            caller.reset_stats()
            caller.reset_site_stats()
        barrier(env)
        t0 = time.perf_counter()
        for it in range(args.steps):
            last_pass = it == args.steps - 1
            before = caller.site_totals() if last_pass else None
            for rc in resident:
                n_windows += genome.walk_contig(caller, rc, args.window, True, stream=stream)
                if last_pass:  # the reference's per-contig copy of the totals (gt_ctg_stats): one 112-byte read per contig
                    after = caller.site_totals()
                    contig_totals[rc.index] = (after - before).astype(np.int64).reshape(-1)
                    before = after
        if gi == len(groups) - 1:
            # the only exchange of a sharded run: the counter block and the statistics block (every field a sum), and the
            # gather of the per-contig totals (each contig is owned by one rank: a sum of zero-padded tables is a gather)
            stats = shard.allreduce_stats(caller.stats_vector(), dev if dist is not None else None)
            site = shard.allreduce_site_stats(caller.site_stats(), dev if dist is not None else None)
            tab = np.zeros((len(lengths), 14), dtype=np.uint64)
            for ci, v in contig_totals.items():
                tab[ci] = v.astype(np.uint64)
            contig_table = shard.allreduce_counts(tab, dev if dist is not None else None)
        barrier(env)
        dt += time.perf_counter() - t0
        if gi == len(groups) - 1:
            # device time of one window, from HIP events around its launches, outside the timed region (and after the
            # counters were read: the extra window is not part of the job)
            caller.set_profiling(True)
            rcl = resident[-1]
            caller.chain_device(rcl.d_cts.data_ptr(), rcl.d_ref.data_ptr(), 1, rcl.length, 0, min(args.window, rcl.length), rcl.d_core.data_ptr(),
                                d_dbsnp=None if rcl.d_dbsnp is None else rcl.d_dbsnp.data_ptr(), with_stats=False, stream=stream)
            chain_ms.append(caller.last_chain_ms())
            caller.set_profiling(False)
            del rcl
        del resident
        torch.cuda.empty_cache()
    dt = max_over_ranks(env, dt)
    # who owned what: the longest-processing-time assignment of whole contigs (shard.assign_contigs), the same on every rank
    share_table = [{"rank": r, "contigs": genome.rank_contigs(lengths, r, v_world), "positions": sum(lengths[c] for c in genome.rank_contigs(lengths, r, v_world))}
                   for r in range(v_world)]
    largest = max(sh["positions"] for sh in share_table)
    for sh in share_table:
        sh["of_largest"] = round(sh["positions"] / largest, 4)
    total = int(stats[0])
    expect = (sum(lengths) if not args.rank_of else my_positions) * args.steps
    assert total == expect, (total, expect)
    if rank != 0:
        caller.close()
        return None
    value = total / dt
    res = {
        "metric": "genome positions called/sec",
        "value": value,
        "unit": "positions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "gbases_per_hour": value * 3.6e-6,
        "rank_time": rank_time_summary(env),  # max / mean over the ranks: the job is as fast as its largest share
        "config": {
            "shares": share_table,
            "workload": ("configs[4]: configs[2] with a dbSNP index loaded (%d sites of this share; synthetic, 1 site / 300 bp, 10 %% "
                         "flagged fq_mask: forced hom-ref records, dbSNP statistics) — " % n_db if args.dbsnp else "")
            + "configs[2]: synthetic human-scale genome, 24 contigs, %d positions at %dx WGBS with 1 %% N-runs "
            "(L-pileup generator), contigs assigned to ranks by longest-processing-time, each contig HBM-resident and walked in "
            "%d-position windows through the fused chain (pile-up -> call -> VCF record -> site statistics)"
            % (sum(lengths), args.coverage, args.window),
            "share": "rank %d of %d%s: contigs %s, %d positions in %d resident group(s)"
            % (v_rank, v_world, " (emulated on one process)" if args.rank_of else "", mine, my_positions, len(groups)),
            "windows_per_step": n_windows // max(args.steps, 1),
            "coverage": args.coverage,
            "sharding": "whole contigs per rank (LPT), no data-path collective; at the end one all-reduce of the 13-word counter "
            "block, one of the statistics block (27 k words) and the gather of the per-contig totals (24 x 14 words)",
            "contigs_with_records_after_gather": int((contig_table[:, 0] > 0).sum()),
            "per_contig_records_sum_equals_total": bool(int(contig_table[:, 0].sum()) * max(args.steps, 1) == int(site["snps"][0])) if not args.rank_of else None,
            "covered_fraction": float(stats[1]) / total,
            "records_written": int(site["snps"][0]) // max(args.steps, 1),
            "CpGs": int(site["CpG_ref"][0] + site["CpG_nonref"][0]) // max(args.steps, 1),
            "dbSNP_sites_written": int(site["dbSNP_sites"][0]) // max(args.steps, 1),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "bsc_chain_kernel_t (bsc_chain_device), all windows of a step",
            "achieved": total * CHAIN_BYTES / dt / 1e9 / (1 if args.rank_of else world),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": total * CHAIN_BYTES / dt / 1e9 / (1 if args.rank_of else world) / HBM_PEAK_GBPS,
            "traffic": None,
            "algorithmic_bytes_per_position": CHAIN_BYTES,
            "one_window_device_ms": chain_ms[-1],
            "note": "per-GPU, from the wall time of the step (windows back to back on one stream); instruction-issue bound, not HBM",
        },
    }
    caller.close()
    return res


def dbsnp_index_for(args, contigs, lengths, rank):
    """configs[4]: write the synthetic index of this rank's contigs (tools/make_dbsnp_index.py: the reference's on-disk
    format) into a temporary file and open it with the library's reader.  Untimed set-up."""
    import importlib.util
    import tempfile

    from bs_call_amd.dbsnp import DbSnpIndex

    spec = importlib.util.spec_from_file_location("make_dbsnp_index", os.path.join(ROOT, "tools", "make_dbsnp_index.py"))
    W = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(W)
    path = os.path.join(tempfile.mkdtemp(prefix="bsc_dbsnp_"), "rank%d.idx" % rank)
    first, ctgs = 1000, {}
    for c in contigs:
        ctgs["ctg%d" % c] = W.synthetic_sites(lengths[c], 300, first_rs=first)
        first += 10 * len(ctgs["ctg%d" % c])
    W.write_index(path, ctgs)
    return DbSnpIndex(path)


if __name__ == "__main__":
    main()
