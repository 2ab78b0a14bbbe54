#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_bench.sh: HBM bytes per bsc_call_kernel launch =
(2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts half of wide coalesced reads, MI355X_MICROARCH.md), tagged
with the hash of the kernel sources so that bench.py stops quoting it once the kernel has changed.
With gpurun_out/prof_<tag>/chain_fetch and chain_write (PMC passes of tools/bench_chain.py, see tools/prof_final.sh) the fused
chain kernel gets an entry of its own ("chain"), tagged with the hash of ITS sources.
usage: python tools/make_traffic_json.py gpurun_out/prof_<tag> [positions] [coverage]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

d = sys.argv[1]
positions = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
coverage = int(sys.argv[3]) if len(sys.argv) > 3 else 30


def per_dispatch(sub, name, kernel="bsc_call_kernel"):
    agg = {}
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name and kernel in r.get("Kernel_Name", "") and "ILb0E" not in r["Kernel_Name"] and "<false" not in r["Kernel_Name"]:
                k = r["Dispatch_Id"]
                agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    v = list(agg.values())
    return sum(v) / len(v)


fetch, write = per_dispatch("pmc_fetch", "FETCH_SIZE"), per_dispatch("pmc_write", "WRITE_SIZE")
out = {
    "_source": "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python bench.py --steps 2 --warmup 1`, bsc_call_kernel, "
    "per dispatch; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads)" % d,
    "positions": positions,
    "coverage": coverage,
    "fetch_size_kib": fetch,
    "write_size_kib": write,
    "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
    "kernel_source_sha256_16": bench.kernel_source_hash(),
}
if os.path.isdir(os.path.join(d, "chain_fetch")):
    cf, cw = per_dispatch("chain_fetch", "FETCH_SIZE", "bsc_chain_kernel"), per_dispatch("chain_write", "WRITE_SIZE", "bsc_chain_kernel")
    out["chain"] = {
        "_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python tools/bench_chain.py --no-unfused --steps 2`, "
        "bsc_chain_kernel_t<true>, per dispatch, with statistics",
        "positions": positions,
        "coverage": coverage,
        "fetch_size_kib": cf,
        "write_size_kib": cw,
        "hbm_bytes_per_launch": int((2 * cf + cw) * 1024),
        "kernel_source_sha256_16": bench.kernel_source_hash(bench.CHAIN_SOURCES),
    }
if os.path.isdir(os.path.join(d, "reads_fetch")):
    # the reads-in chain's kernel (READS = true) and the stand-alone accumulate kernel, from tools/bench_reads.py under --pmc
    rf, rw = per_dispatch("reads_fetch", "FETCH_SIZE", "bsc_chain_kernel_t<true, true>"), per_dispatch("reads_write", "WRITE_SIZE", "bsc_chain_kernel_t<true, true>")
    af, aw = per_dispatch("reads_fetch", "FETCH_SIZE", "bsc_accumulate_kernel"), per_dispatch("reads_write", "WRITE_SIZE", "bsc_accumulate_kernel")
    src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python tools/bench_reads.py --steps 2 --no-check`, per dispatch"
    h = bench.kernel_source_hash(bench.READS_SOURCES)
    out["reads"] = {"_source": src + ", bsc_chain_kernel_t<true, true> (the dominant kernel of bsc_reads_chain_device)", "positions": positions,
                    "coverage": coverage, "fetch_size_kib": rf, "write_size_kib": rw, "hbm_bytes_per_launch": int((2 * rf + rw) * 1024),
                    "kernel_source_sha256_16": h}
    out["accumulate"] = {"_source": src + ", bsc_accumulate_kernel (the dominant kernel of bsc_accumulate_device)", "positions": positions,
                         "coverage": coverage, "fetch_size_kib": af, "write_size_kib": aw, "hbm_bytes_per_launch": int((2 * af + aw) * 1024),
                         "kernel_source_sha256_16": h}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out))
