#!/usr/bin/env python3
"""File to file at BASELINE configs[1]'s size: a synthetic coordinate-sorted WGBS BAM (tools/make_wgbs_bam.c) + FASTA -> BCF + report through
integration/bam2bcf, with the reader on the device (round 6: the host only inflates) and, for the same bytes, with the host reader of rounds 2-5.
Also the host streamer alone (its inflate rate by helper count) and the device reader alone (blocks formed, nothing called).
usage: python tools/bench_bam2bcf_big.py [positions [coverage [out.json [contigs [straddle [poisson]]]]]]     (writes its files under /tmp)"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
out_json = sys.argv[3] if len(sys.argv) > 3 else None
contigs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
straddle = int(sys.argv[5]) if len(sys.argv) > 5 else 0
poisson = int(sys.argv[6]) if len(sys.argv) > 6 else 0  # forward starts as a Poisson process: coverage gaps, hence many blocks, at low depth
d = os.environ.get("BENCH_TMP", "/tmp/bam2bcf_big")
os.makedirs(d, exist_ok=True)
gen = os.path.join(d, "make_wgbs_bam")
subprocess.check_call(["gcc", "-O2", "-o", gen, os.path.join(ROOT, "tools", "make_wgbs_bam.c"), "-lz", "-lpthread", "-lm"])
bam, fa = os.path.join(d, "in.bam"), os.path.join(d, "ref.fa")
t0 = time.time()
ginfo = json.loads(subprocess.check_output([gen, bam, fa, str(n), str(cov), "88172645463325253", str(min(32, os.cpu_count() or 8)), "1", str(straddle), str(contigs), str(poisson)]))
gen_s = time.time() - t0
print("generated", ginfo, round(gen_s, 1), "s", os.path.getsize(bam), "bytes", flush=True)
res = {"positions": n, "coverage": cov, "contigs": contigs, "records_straddle_bgzf_blocks": bool(straddle), "poisson_starts": bool(poisson), "alignments": ginfo["alignments"],
       "bam_bytes": os.path.getsize(bam), "generate_s": round(gen_s, 1), "host_cpus": os.cpu_count(),
       "host_cpus_usable": len(os.sched_getaffinity(0))}

from bs_call_amd.bamdev import drain_stream  # noqa: E402

res["host_streamer_alone"] = []
for th in (8, 16, 32, 64, 96, 0):
    nb, nr, dt, nth = drain_stream(bam, threads=th)
    res["host_streamer_alone"].append({"helper_threads": nth, "seconds": round(dt, 3), "inflated_GB_per_s": round(nb / dt / 1e9, 3), "positions_per_s": round(n / dt)})
    res["inflated_bytes"] = nb
    print("streamer", res["host_streamer_alone"][-1], flush=True)

if os.environ.get("BENCH_ONLY_STREAMER"):
    os.environ["BSC_BAMSTREAM_NOWALK"] = "1"
    for th in (16, 32, 64):
        nb, nr, dt, nth = drain_stream(bam, threads=th)
        print("nowalk", nth, round(dt, 3), round(nb / dt / 1e9, 3), "GB/s", flush=True)
    del os.environ["BSC_BAMSTREAM_NOWALK"]
    for th in (16, 32, 64):
        nb, nr, dt, nth = drain_stream(bam, threads=th, slab_bytes=8 << 20, n_slabs=32)
        print("8MB slabs x32", nth, round(dt, 3), round(nb / dt / 1e9, 3), "GB/s", flush=True)
    for th in (16, 32, 64):
        nb, nr, dt, nth = drain_stream(bam, threads=th, slab_bytes=64 << 20, n_slabs=16)
        print("64MB slabs x16", nth, round(dt, 3), round(nb / dt / 1e9, 3), "GB/s", flush=True)
    print(json.dumps(res))
    raise SystemExit(0)
exe = os.path.join(ROOT, "bs_call_amd", "lib", "bam2bcf")
runs = {}
for mode, env_extra, reps in (("device_reader", {}, 3), ("host_reader", {"BAM2BCF_HOST_READER": "1", "BAM2BCF_THREADS": "4"}, 1)):
    best = None
    for _ in range(reps):
        env = dict(os.environ, BAM2BCF_TIMING="1", **env_extra)
        ob, orp = os.path.join(d, mode + ".bcf"), os.path.join(d, mode + ".json")
        for f_ in (ob, orp):  # (removing gigabytes of an earlier run's page cache is not this run's work)
            if os.path.exists(f_):
                os.remove(f_)
        t0 = time.time()
        r = subprocess.run([exe, bam, fa, ob, orp], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        if r.returncode != 0:
            print(r.stderr[-2000:])
            raise SystemExit(1)
        stages = json.loads(r.stderr.strip().splitlines()[-1])
        if best is None or dt < best[0]:
            best = (dt, stages, r.stdout.strip())
        print(mode, round(dt, 3), stages, flush=True)

    def sha(p):
        h = hashlib.sha256()
        with open(p, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                h.update(chunk)
        return h.hexdigest()

    runs[mode] = {"process_wall_s_best": round(best[0], 3), "stages": best[1], "stdout": best[2], "bcf_sha256": sha(ob), "report_sha256": sha(orp),
                  "bcf_bytes": os.path.getsize(ob)}
res["bam2bcf"] = runs
res["same_bcf_and_report_bytes"] = runs["device_reader"]["bcf_sha256"] == runs["host_reader"]["bcf_sha256"] and runs["device_reader"]["report_sha256"] == runs["host_reader"]["report_sha256"]
w = runs["device_reader"]["stages"]["wall_without_context_s"]
res["device_reader_positions_per_s_without_context"] = round(n / w)
res["device_reader_positions_per_s_whole_process"] = round(n / runs["device_reader"]["process_wall_s_best"])
res["host_reader_positions_per_s_whole_process"] = round(n / runs["host_reader"]["process_wall_s_best"])
print(json.dumps(res))
if out_json:
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
assert res["same_bcf_and_report_bytes"], "the device reader's run wrote other bytes than the host reader's"
