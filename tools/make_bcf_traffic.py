#!/usr/bin/env python3
"""profiles/traffic.json["bcf"] from the PMC passes of tools/pmc_bcf.sh: HBM bytes per launch of the BCF encoder's two kernels =
(2 * FETCH_SIZE + WRITE_SIZE) KiB (the gfx950 rule of MI355X_MICROARCH.md, as tools/make_traffic_json.py applies it), per record, tagged
with the hash of csrc/bcfdev.hip.  usage: python tools/make_bcf_traffic.py gpurun_out/<tag>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

d = sys.argv[1]


def per_dispatch(counter, kernel):
    agg = {}
    for f in glob.glob(os.path.join(d, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and kernel in r.get("Kernel_Name", ""):
                agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = list(agg.values())
    return sum(v) / len(v)


run = json.loads(open(os.path.join(d, "pmc_FETCH_SIZE.json")).read().strip().splitlines()[-1])
n_rec, nbytes = run["records"], run["bcf_bytes"]
ks = {}
for k in ("bsc_bcf_size_kernel", "bsc_bcf_write_kernel"):
    f, w = per_dispatch("FETCH_SIZE", k), per_dispatch("WRITE_SIZE", k)
    ks[k] = {"fetch_size_kib": f, "write_size_kib": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
total = sum(v["hbm_bytes_per_launch"] for v in ks.values())
alg = 2 * 128 * n_rec + nbytes
entry = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python tools/bench_bcf.py --sites 50000000 --steps 2` (tools/pmc_bcf.sh), "
         "bsc_bcf_size_kernel + bsc_bcf_write_kernel (bsc_bcf_block_device over packed records), per dispatch",
         "records": n_rec, "bcf_bytes": nbytes, "kernels": ks, "hbm_bytes_per_launch": total, "hbm_bytes_per_record": total / n_rec,
         "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": total / alg,
         "kernel_source_sha256_16": bench.kernel_source_hash(("bcfdev.hip",))}
p = os.path.join(ROOT, "profiles", "traffic.json")
out = json.load(open(p))
out["bcf"] = entry
json.dump(out, open(p, "w"), indent=1)
print("# BCF encoder, %d records (50 M positions at 30x), HBM bytes per launch from FETCH_SIZE x 2 + WRITE_SIZE (KiB)" % n_rec)
for k, v in ks.items():
    print("%-24s fetch %12.0f KiB  write %12.0f KiB  -> %.3f GB" % (k, v["fetch_size_kib"], v["write_size_kib"], v["hbm_bytes_per_launch"] / 1e9))
print("total %.3f GB = %.1f B per record; algorithmic %.3f GB (2 x 128 B + %.1f B per record): %.3f x" % (total / 1e9, total / n_rec, alg / 1e9, nbytes / n_rec, total / alg))
