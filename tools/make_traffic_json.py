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
            kn = r.get("Kernel_Name", "")
            # the launch's complete tiles only (bsc_call_kernel_t<true>, bsc_chain_kernel_t<true, ...>), unless a full name was asked for
            if r.get("Counter_Name") == name and kernel in kn and ("<" in kernel or ("ILb0E" not in kn and "_t<false" not in kn)):
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
    cf, cw = per_dispatch("chain_fetch", "FETCH_SIZE", "bsc_chain_kernel_t<true, false, false, false>"), per_dispatch("chain_write", "WRITE_SIZE", "bsc_chain_kernel_t<true, false, false, false>")
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
    # tools/bench_reads.py under --pmc: the stand-alone accumulate kernel (bsc_accumulate_device) and the two kernels of
    # bsc_reads_chain_device's default form — the accumulate kernel's summary form and the chain kernel's summary-in form
    def two(kernel):
        return per_dispatch("reads_fetch", "FETCH_SIZE", kernel), per_dispatch("reads_write", "WRITE_SIZE", kernel)

    af, aw = two("bsc_accumulate_kernel_t<false>")
    sf, sw = two("bsc_accumulate_kernel_t<true>")
    cf, cw = two("bsc_chain_kernel_t<true, false, false, true>")
    src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python tools/bench_reads.py --steps 2 --no-check`, per dispatch"
    h = bench.kernel_source_hash(bench.READS_SOURCES)
    out["reads"] = {"_source": src + ", bsc_accumulate_kernel_t<true> + bsc_chain_kernel_t<true, false, false, true> (the two kernels of "
                    "bsc_reads_chain_device; the 48-byte summaries are written by the first and read by the second: HBM bytes the "
                    "algorithmic count does not have)", "positions": positions, "coverage": coverage,
                    "fetch_size_kib": sf + cf, "write_size_kib": sw + cw, "hbm_bytes_per_launch": int((2 * (sf + cf) + sw + cw) * 1024),
                    "kernels": {"bsc_accumulate_kernel_t<true>": {"fetch_size_kib": sf, "write_size_kib": sw},
                                "bsc_chain_kernel_t<true, false, false, true>": {"fetch_size_kib": cf, "write_size_kib": cw}},
                    "kernel_source_sha256_16": h}
    out["accumulate"] = {"_source": src + ", bsc_accumulate_kernel_t<false> (the dominant kernel of bsc_accumulate_device)", "positions": positions,
                         "coverage": coverage, "fetch_size_kib": af, "write_size_kib": aw, "hbm_bytes_per_launch": int((2 * af + aw) * 1024),
                         "kernel_source_sha256_16": h}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out))
