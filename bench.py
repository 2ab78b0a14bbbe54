#!/usr/bin/env python3
"""bench.py — genome positions called per second by the gfx950 calling path (pile-up -> gt_meth).

Contract (see the task prompt):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
  * a "step" = one pass of the hot path over this rank's synthetic contig (config 2 of BASELINE.json:
    chr22-sized, 50 Mb at 30x), pile-ups and reference codes already resident in HBM (generated on the
    device by bsc_synth_pileup_device); the timed region covers the calling kernel, the Fisher pass and,
    once at the end, the RCCL all-reduce of the per-rank counters.
  * N > 1: one process per GPU (torch.distributed / RCCL); contigs shard across ranks with no data-path
    collective (weak scaling: every rank calls its own 50 Mb contig), value = all sites / max-over-ranks time.
  * roofline: achieved = 305 algorithmic bytes/site (104 B pileup + 1 B ref + 200 B gt_meth)
    x sites per launch / average device time of the calling kernel, measured with HIP events recorded on
    the launch stream inside libbscall_amd (bsc_set_profiling / bsc_last_kernel_ms).
  * cpu_baseline: the CPU oracle (libm flavour = restatement of the reference; "port") timed on this host's
    cores over the first --cpu-sites positions of the same synthetic contig (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_COVERED = 305
ALGO_BYTES_UNCOVERED = 105
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 88172645463325252


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sites", type=int, default=50_000_000, help="positions per rank (config 2: 50 Mb)")
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--cpu-sites", type=int, default=32_000_000, help="sample size of the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    import bs_call_amd as B

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
    dev = torch.device("cuda", torch.cuda.current_device())

    n = args.sites
    caller = B.SiteCaller(device=dev.index)
    d_cts = torch.empty(n * 104, dtype=torch.uint8, device=dev)
    d_ref = torch.empty(n, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
    d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # rank r calls contig r of the synthetic genome: same generator, disjoint site range
    first_site = rank * n
    caller.synth_device(SEED + 2, first_site, n, args.coverage, d_cts.data_ptr(), d_ref.data_ptr(), 0, stream)
    torch.cuda.synchronize()

    def step():
        caller.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, stream)

    caller.set_profiling(True)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    caller.reset_stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = []
    fisher_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if world == 1:
            # reading the events waits for this launch only; the next launch is queued right after
            a, b = caller.last_kernel_ms()
            kernel_ms.append(a)
            fisher_ms.append(b)
    stats = torch.from_numpy(caller.stats_vector()).to(dev)  # syncs the stream
    if dist is not None:
        dist.all_reduce(stats)  # the only collective: per-rank counters (RCCL)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if world > 1:
        a, b = caller.last_kernel_ms()
        kernel_ms, fisher_ms = [a], [b]

    stats = stats.cpu().numpy()
    total_sites = int(stats[0])
    covered = int(stats[1])
    assert total_sites == n * world * args.steps, (total_sites, n, world, args.steps)
    value = total_sites / dt

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
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms_avg": k_ms,
                "fisher_kernel_ms_avg": float(np.mean(fisher_ms)),
                "positions_per_s_kernel_only": n / (k_ms * 1e-3),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, d_cts, d_ref, d_out, d_skip)
        # Last (it overwrites d_out): the same bytes moved by a kernel that does nothing else — what the memory system
        # delivers for this 1 : 2 read : write mix (HBM3E writes stream slower than reads), measured live on this GPU.
        probe_ms = caller.stream_probe_ms(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 5, stream)
        if probe_ms > 0.0:
            res["roofline"]["stream_probe"] = {
                "what": "copy kernel with the calling kernel's traffic and tile shape, no arithmetic (csrc/probe.hip), best of 5",
                "ms": probe_ms,
                "GBps": algo_bytes / (probe_ms * 1e-3) / 1e9,
                "kernel_frac_of_probe": probe_ms / k_ms,
            }
        print(json.dumps(res), flush=True)
    caller.close()
    if dist is not None:
        dist.destroy_process_group()


def profiled_traffic(n, coverage):
    """HBM bytes per bsc_call_kernel launch from the committed PMC passes (profiles/traffic.json, written from
    `tools/profile_bench.sh`: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE
    doubled per the gfx950 rule).  Counters cannot be read inside this process; None if the workload differs."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if t.get("positions") == n and t.get("coverage") == coverage:
            return t["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(args, d_cts, d_ref, d_out, d_skip):
    """Time the CPU oracle (libm flavour) on a bounded sample of the same workload and, while at it,
    check the GPU output of that sample against the oracle (bsm flavour, bit-exact)."""
    import numpy as np

    import bs_call_amd as B
    from oracle import loader as O

    m = min(args.cpu_sites, args.sites)
    pile = d_cts[: m * 104].cpu().numpy().view(B.PILEUP)
    ref = d_ref[:m].cpu().numpy()
    tb = O.Tables()
    cores = os.cpu_count() or 1
    O.call_sites(pile[:100_000], ref[:100_000], tb, O.LIBM, -cores)  # warm-up (page in the library)
    out = np.zeros(m, dtype=B.GT_METH)
    skip = np.zeros(m, dtype=np.uint8)
    out[:] = out  # touch the pages so the timing below is compute, not first-touch faults
    L = O.lib()
    t0 = time.perf_counter()
    L.orc_call_sites(pile.ctypes.data, ref.ctypes.data, m, tb.ptr, out.ctypes.data, skip.ctypes.data, O.LIBM, -cores)
    t_all = time.perf_counter() - t0
    m1 = min(m, 4_000_000)
    t0 = time.perf_counter()
    L.orc_call_sites(pile.ctypes.data, ref.ctypes.data, m1, tb.ptr, out.ctypes.data, skip.ctypes.data, O.LIBM, 1)
    t_one = time.perf_counter() - t0
    # parity of the benchmarked output on a slice of the sample (GPU vs bsm flavour: every byte)
    mc = min(m, 1_000_000)
    exp, eskip = O.call_sites(pile[:mc], ref[:mc], tb, O.BSM, -cores)
    got = d_out[: mc * 200].cpu().numpy().view(B.GT_METH)
    ok = got.tobytes() == exp.tobytes() and (d_skip[:mc].cpu().numpy() == eskip).all()
    return {
        "value": m / t_all,
        "unit": "positions/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d positions of the benchmarked contig, oracle libm flavour, %d threads on contiguous ranges; "
        "single thread on %d positions: %.3g positions/s" % (m, cores, m1, m1 / t_one),
        "single_thread_value": m1 / t_one,
        "gpu_output_matches_oracle_on_sample": bool(ok),
    }


if __name__ == "__main__":
    main()
