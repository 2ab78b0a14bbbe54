#!/usr/bin/env python3
"""A/B of library build variants on the pile-up-in chain and the calling kernel: one process per variant.
usage: python tools/ab_chain.py [--sites N] [--coverage C] [names...]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sites, cov, names, i = "50000000", "30", [], 0
while i < len(args):
    if args[i] == "--sites": sites = args[i + 1]; i += 2
    elif args[i] == "--coverage": cov = args[i + 1]; i += 2
    else: names.append(args[i]); i += 1
libs = [("main", os.path.join(ROOT, "bs_call_amd", "lib", "libbscall_amd.so"))]
for f in sorted(glob.glob(os.path.join(ROOT, "bs_call_amd", "lib", "variants", "lib_*.so"))):
    n = os.path.basename(f)[4:-3]
    if not names or n in names:
        libs.append((n, f))
for rep in range(2):
    for n, f in libs:
        env = dict(os.environ, BSCALL_AMD_LIB=f)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--sites", sites, "--coverage", cov, "--steps", "10", "--no-reads"],
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            print("%-28s call %.3f ms (min %.3f)  chain %.3f ms" % (n, d["roofline"]["kernel_ms_avg"], d["roofline"]["kernel_ms_min"],
                  d["roofline_chain"]["kernel_ms_avg"]), flush=True)
        except Exception as e:
            print(n, "FAILED", p.stderr[-400:], flush=True)
