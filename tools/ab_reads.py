#!/usr/bin/env python3
"""A/B of library build variants on the reads-in chain and the accumulate stage (tools/build_variant_fused.sh <name> <flags>):
runs tools/bench_reads.py once per variant, each in its own process.  usage: python tools/ab_reads.py [--sites N] [--coverage C] [names...]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sites, cov = "20000000", "30"
names = []
i = 0
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
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_reads.py"), "--sites", sites, "--coverage", cov, "--steps", "6", "--no-check"],
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            print("%-28s acc %.3f ms   reads_chain %.3f ms (min %.3f)   %.2f G pos/s" % (n, d["accumulate"]["device_ms_avg"], d["reads_chain"]["device_ms_avg"],
                  d["reads_chain"]["device_ms_min"], d["reads_chain"]["G_positions_per_s"]), flush=True)
        except Exception as e:
            print(n, "FAILED", p.stderr[-400:], flush=True)
