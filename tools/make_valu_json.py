#!/usr/bin/env python3
"""profiles/valu.json from the SQ counter passes (tools/pmc_sq.sh, pmc_chain.sh, pmc_kernel.sh): VALU wave-instructions per
launch of every kernel of a leg of bench.py, and the clock the dominant kernel ran at (SQ_BUSY_CYCLES summed over the 32
shader engines / 32 / its duration in the same pass), tagged with the hash of the kernel sources as profiles/traffic.json is —
bench.py turns them into the fraction of the VALU issue capacity (1 024 SIMDs, one wave-instruction per 4 cycles) a leg uses.
usage: python tools/make_valu_json.py [--call DIR] [--chain DIR] [--reads DIR] [--acc DIR] [--positions N] [--coverage C]
       (DIR: the directory holding the passes' p1/ p2/ ... subdirectories)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

args = sys.argv[1:]
opt = {args[i][2:]: args[i + 1] for i in range(0, len(args), 2)}
positions, coverage = int(opt.get("positions", 50_000_000)), int(opt.get("coverage", 30))
SE = 32  # shader engines: SQ_BUSY_CYCLES is summed over them


def per_kernel(d):
    """kernel name -> {counter: mean per dispatch}, 'ns': mean duration, 'n': dispatches"""
    agg, dur = {}, {}
    for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], f, r["Dispatch_Id"])
            agg.setdefault(k, {})
            agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for (name, f, _), c in agg.items():
        o = out.setdefault(name, {"_n": {}, "_sum": {}})
        for cn, v in c.items():
            o["_sum"][cn] = o["_sum"].get(cn, 0.0) + v
            o["_n"][cn] = o["_n"].get(cn, 0) + 1
        if "SQ_BUSY_CYCLES" in c:  # the duration of the dispatches that carry the clock counter
            o["_sum"]["ns"] = o["_sum"].get("ns", 0.0) + dur[(name, f, _)]
            o["_n"]["ns"] = o["_n"].get("ns", 0) + 1
    return {n: {cn: o["_sum"][cn] / o["_n"][cn] for cn in o["_sum"]} for n, o in out.items()}


def leg(d, parts, sources):
    """parts: [(kernel-name substring, launches per leg, must-not-contain)], the first one dominant"""
    k = per_kernel(d)
    kernels, total, clock = {}, 0.0, None
    for i, (sub, mult, *excl) in enumerate(parts):
        hit = [n for n in k if sub in n and not any(e in n for e in excl) and "SQ_INSTS_VALU" in k[n]]
        if not hit:
            if i == 0:
                raise SystemExit("no %s in %s" % (sub, d))
            continue
        n = hit[0]
        kernels[sub] = {"insts_valu_per_launch": k[n]["SQ_INSTS_VALU"], "launches_per_leg": mult}
        total += k[n]["SQ_INSTS_VALU"] * mult
        if i == 0 and "SQ_BUSY_CYCLES" in k[n]:
            clock = k[n]["SQ_BUSY_CYCLES"] / SE / k[n]["ns"]
            kernels[sub].update(sq_busy_cycles=k[n]["SQ_BUSY_CYCLES"], ns_in_that_pass=k[n]["ns"])
    return {"_source": "%s: rocprofv3 --pmc passes (SQ_INSTS_VALU; SQ_BUSY_CYCLES / %d shader engines / duration = clock)" % (d, SE),
            "positions": positions, "coverage": coverage, "insts_valu_per_launch": total, "clock_ghz": clock, "kernels": kernels,
            "kernel_source_sha256_16": bench.kernel_source_hash(sources)}


path = os.path.join(ROOT, "profiles", "valu.json")
out = json.load(open(path)) if os.path.exists(path) else {}
if "call" in opt:
    out["call"] = leg(opt["call"], [("bsc_call_kernel", 1), ("bsc_fisher_kernel", 1)], bench.KERNEL_SOURCES)
# kernel-name substrings as rocprofv3 prints them: bsc_chain_kernel_t<FULL, READS, MULTI, SUMM>, bsc_accumulate_kernel_t<SUMM>
if "chain" in opt:
    out["chain"] = leg(opt["chain"], [("bsc_chain_kernel_t<true, false, false, false>", 1), ("bsc_chain_kernel_t<false, false, false, false>", 2)],
                       bench.CHAIN_SOURCES)
if "reads" in opt:  # bsc_reads_chain_device's default form: summaries through HBM
    out["reads"] = leg(opt["reads"], [("bsc_chain_kernel_t<true, false, false, true>", 1), ("bsc_chain_kernel_t<false, false, false, true>", 2),
                                      ("bsc_accumulate_kernel_t<true>", 1), ("bsc_bin_count_kernel", 1), ("bsc_bin_scatter_kernel", 1)],
                       bench.READS_SOURCES)
if "acc" in opt:
    out["accumulate"] = leg(opt["acc"], [("bsc_accumulate_kernel_t<false>", 1), ("bsc_bin_count_kernel", 1), ("bsc_bin_scatter_kernel", 1)],
                            bench.READS_SOURCES)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
