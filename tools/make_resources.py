#!/usr/bin/env python3
"""profiles/r06_resources.txt: per kernel of the library, what the compiler made of HEAD's sources — VGPRs, spilled VGPRs / SGPRs,
scratch bytes per lane, LDS bytes per workgroup (-Rpass-analysis=kernel-resource-usage), code bytes, and the static counts of
v_readlane / v_writelane / scratch_* instructions in the listing.  Every resource figure quoted in DESIGN.md comes from this
file; regenerate it after touching a kernel.  usage: python tools/make_resources.py [out-file]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_resources.txt")
FLAGS = "-O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wno-inline-asm -Wno-unused-variable -Wno-unused-function".split()
SOURCES = ["kernels.hip", "fused.hip", "accumulate.hip", "vcfcore.hip", "sitestats.hip", "compact.hip", "prepdev.hip", "bcfdev.hip", "bamdev.hip"]


def demangle(n):
    try:
        return re.sub(r"^void ", "", subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0])
    except Exception:
        return n


rows = []
for src in SOURCES:
    path = os.path.join(ROOT, "bs_call_amd", "csrc", src)
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        p = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", path, "-o", asm], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        res, cur = {}, None
        for ln in p.stderr.splitlines():
            m = re.search(r"remark: (?:\s*)Function Name: (\S+)", ln)
            if m:
                cur = m.group(1)
                res[cur] = {}
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
            if m and cur:
                res[cur][m.group(1).strip()] = int(m.group(2))
        # static instruction counts and code size per function from the listing
        text, name = open(asm).read().splitlines(), None
        counts = {}
        for ln in text:
            s = ln.strip()
            m = re.match(r"([A-Za-z_][\w$.]*):\s*(;.*)?$", s)
            if m and m.group(1) in res:
                name = m.group(1)
                counts[name] = {"readlane": 0, "writelane": 0, "scratch_load": 0, "scratch_store": 0, "insts": 0}
                continue
            if s.startswith(".Lfunc_end"):
                name = None
            if name and s and not s.startswith((".", ";")) and not re.match(r"[\w$.]+:", s):
                op = s.split()[0]
                c = counts[name]
                c["insts"] += 1
                if op.startswith("v_readlane"):
                    c["readlane"] += 1
                elif op.startswith("v_writelane"):
                    c["writelane"] += 1
                elif op.startswith("scratch_load"):
                    c["scratch_load"] += 1
                elif op.startswith("scratch_store"):
                    c["scratch_store"] += 1
        for k, r in res.items():
            if "VGPRs" not in r:
                continue
            c = counts.get(k, {})
            rows.append((src, demangle(k), r.get("VGPRs", 0), r.get("VGPRs Spill", 0), r.get("SGPRs Spill", 0), r.get("ScratchSize", r.get("ScratchSize [bytes/lane]", 0)),
                         r.get("LDS Size", r.get("LDS Size [bytes/block]", 0)), r.get("Occupancy", r.get("Occupancy [waves/SIMD]", 0)), c.get("insts", 0),
                         c.get("readlane", 0), c.get("writelane", 0), c.get("scratch_load", 0), c.get("scratch_store", 0)))
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
with open(out_path, "w") as f:
    f.write("# tools/make_resources.py at %s (+ working tree): hipcc %s, per kernel\n" % (head, " ".join(FLAGS[:6])))
    f.write("# bsc_chain_kernel_t<FULL, READS, MULTI, SUMM>; bsc_accumulate_kernel_t<SUMM>; bsc_call_kernel_t<FULL>; static counts = instructions in the listing\n")
    f.write("%-16s %-62s %5s %7s %7s %8s %8s %5s %7s %9s %10s %8s %8s\n" % ("source", "kernel", "VGPRs", "spillV", "spillS", "scratchB", "LDS B", "occ", "insts", "readlane", "writelane", "scr_ld", "scr_st"))
    for r in sorted(rows):
        f.write("%-16s %-62s %5d %7d %7d %8d %8d %5d %7d %9d %10d %8d %8d\n" % r)
print(open(out_path).read())
