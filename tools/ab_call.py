#!/usr/bin/env python3
"""A/B of library build variants on the calling kernel alone, over several launch sizes: one process per variant and size.
usage: python tools/ab_call.py [--coverage C] [--sizes a,b,c] [names...]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sizes, cov, names, i = "1000000,4000000,10000000,50000000", "30", [], 0
while i < len(args):
    if args[i] == "--sizes": sizes = args[i + 1]; i += 2
    elif args[i] == "--coverage": cov = args[i + 1]; i += 2
    else: names.append(args[i]); i += 1
libs = [("main", os.path.join(ROOT, "bs_call_amd", "lib", "libbscall_amd.so"))]
for f in sorted(glob.glob(os.path.join(ROOT, "bs_call_amd", "lib", "variants", "lib_*.so"))):
    n = os.path.basename(f)[4:-3]
    if not names or n in names:
        libs.append((n, f))
for sites in sizes.split(","):
    for rep in range(2):
        for n, f in libs:
            env = dict(os.environ, BSCALL_AMD_LIB=f)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--sites", sites, "--coverage", cov, "--steps", "20",
                                "--no-reads", "--no-chain"], env=env, capture_output=True, text=True)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
                r = d["roofline"]
                print("%9s %-12s call %.4f ms (min %.4f) frac %.4f  probe %.4f" % (sites, n, r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], r["stream_probe"]["ms"]), flush=True)
            except Exception as e:
                print(n, "FAILED", p.stderr[-400:], flush=True)
