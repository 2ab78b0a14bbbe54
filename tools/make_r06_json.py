#!/usr/bin/env python3
"""profiles/traffic.json and profiles/valu.json entries of round 6's legs of bench.py, from the PMC passes of tools/prof_r06.sh:
  prep / prep_profile   tools/bench_prep.py [--profile]: every kernel of bsc_prepare_templates_device (plan, the prefix sums, copy; with the
                        read profile also the max-scan, bsc_prep_refmask_kernel, bsc_prep_profile_kernel), per call
  bcf_sites             tools/bench_sites_bcf.py: bsc_bcf_size_bytes_kernel + the prefix sum + bsc_bcf_write_kernel over the chain's arrays
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB per dispatch (the gfx950 rule of MI355X_MICROARCH.md), summed over the kernels of one call;
VALU wave-instructions = SQ_INSTS_VALU summed the same way.  Each entry carries the hash of its kernel source.
usage: python tools/make_r06_json.py gpurun_out/<tag>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

d = sys.argv[1]


def per_call(sub, counter, calls):
    """{kernel: total of `counter` over its dispatches / calls}: what one call of the entry costs"""
    agg = {}
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                kn = r["Kernel_Name"].split("(")[0]
                kn = kn[5:] if kn.startswith("void ") else kn  # a template instance's name: "void bsc_bcf_write_kernel_t<true, 8192u, 4u>"
                kn = "rocprim scan" if "rocprim" in kn.lower() else kn
                agg[kn] = agg.get(kn, 0.0) + float(r["Counter_Value"])
    return {k: v / calls for k, v in agg.items()}


def last_json(path):
    return json.loads([l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1])


tp, vp = os.path.join(ROOT, "profiles", "traffic.json"), os.path.join(ROOT, "profiles", "valu.json")
traffic, valu = json.load(open(tp)), json.load(open(vp))
for key, tag in (("prep", "prep"), ("prep_profile", "prepp")):
    if not os.path.isdir(os.path.join(d, tag + "_FETCH_SIZE")):
        continue
    run = last_json(os.path.join(d, tag + "_FETCH_SIZE.out"))
    calls = run["calls"]
    keep = lambda m: {k: v for k, v in m.items() if k.startswith("bsc_prep") or k == "rocprim scan"}
    f, w = keep(per_call(tag + "_FETCH_SIZE", "FETCH_SIZE", calls)), keep(per_call(tag + "_WRITE_SIZE", "WRITE_SIZE", calls))
    total = int(sum((2 * f.get(k, 0) + w.get(k, 0)) * 1024 for k in set(f) | set(w)))
    alg = run["bytes_in_plus_out"]
    traffic[key] = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python tools/bench_prep.py %s--steps 2` (tools/prof_r06.sh), every kernel of "
                    "bsc_prepare_templates_device, per call" % ("--profile " if key == "prep_profile" else ""),
                    "sites": run["positions"], "coverage": run["coverage"], "hbm_bytes_per_launch": total, "algorithmic_bytes_per_launch": alg,
                    "traffic_over_algorithmic": total / alg,
                    "kernels": {k: {"fetch_size_kib": f.get(k, 0.0), "write_size_kib": w.get(k, 0.0)} for k in sorted(set(f) | set(w))},
                    "kernel_source_sha256_16": bench.kernel_source_hash(("prepdev.hip",))}
    if os.path.isdir(os.path.join(d, tag + "_SQ")):
        run = last_json(os.path.join(d, tag + "_SQ.out"))
        v = keep(per_call(tag + "_SQ", "SQ_INSTS_VALU", run["calls"]))
        valu[key] = {"_source": "rocprofv3 --pmc SQ_INSTS_VALU pass of the same command", "sites": run["positions"], "coverage": run["coverage"],
                     "valu_insts_per_launch": sum(v.values()), "kernels": v, "kernel_source_sha256_16": bench.kernel_source_hash(("prepdev.hip",))}
if os.path.isdir(os.path.join(d, "sites_FETCH_SIZE")):
    run = last_json(os.path.join(d, "sites_FETCH_SIZE.out"))
    calls = 2 + 2  # the tool's two untimed calls + --steps 2
    keep = lambda m: {k: v for k, v in m.items() if k.startswith("bsc_bcf") or k == "rocprim scan"}
    f, w = keep(per_call("sites_FETCH_SIZE", "FETCH_SIZE", calls)), keep(per_call("sites_WRITE_SIZE", "WRITE_SIZE", calls))
    total = int(sum((2 * f.get(k, 0) + w.get(k, 0)) * 1024 for k in set(f) | set(w)))
    traffic["bcf_sites"] = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python tools/bench_sites_bcf.py --steps 2` (tools/prof_r06.sh): "
                            "bsc_bcf_size_bytes_kernel + the prefix sum + bsc_bcf_write_kernel over the chain's per-position arrays, per call",
                            "sites": run["sites"], "records": run["records"], "bcf_bytes": run["bcf_bytes"], "hbm_bytes_per_launch": total,
                            "hbm_bytes_per_position": total / run["sites"], "algorithmic_bytes_per_launch": run["algorithmic_bytes"],
                            "traffic_over_algorithmic": total / run["algorithmic_bytes"],
                            "kernels": {k: {"fetch_size_kib": f.get(k, 0.0), "write_size_kib": w.get(k, 0.0)} for k in sorted(set(f) | set(w))},
                            "kernel_source_sha256_16": bench.kernel_source_hash(("bcfdev.hip",))}
json.dump(traffic, open(tp, "w"), indent=1)
json.dump(valu, open(vp, "w"), indent=1)
for k in ("prep", "prep_profile", "bcf_sites"):
    if k in traffic:
        t = traffic[k]
        print("%-13s HBM %.3f GB per call = %.3f x algorithmic (%.3f GB)%s" % (k, t["hbm_bytes_per_launch"] / 1e9, t["traffic_over_algorithmic"], t["algorithmic_bytes_per_launch"] / 1e9,
                                                                            "; VALU %.4g wave-instructions" % valu[k]["valu_insts_per_launch"] if k in valu else ""))
        for kn, v in t["kernels"].items():
            print("    %-44s fetch %12.0f KiB  write %12.0f KiB" % (kn[-44:], v["fetch_size_kib"], v["write_size_kib"]))
