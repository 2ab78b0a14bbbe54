#!/usr/bin/env python3
"""A/B kernel build variants in ONE process-per-variant loop, interleaved rounds (programming guide rule 24).
usage: python tools/ab_variants.py [--sites N] [--rounds R] name1 name2 ...   (names under bs_call_amd/lib/variants)
Each variant runs in its own subprocess per round (the library is chosen at load time); prints median/min kernel ms."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sites, rounds, chain, coverage = 50_000_000, 3, False, 30
while args and args[0].startswith("--"):
    if args[0] == "--chain":  # time the fused chain (tools/bench_chain.py) instead of the calling kernel (bench.py)
        chain = True
        args = args[1:]
        continue
    if args[0] == "--sites":
        sites = int(args[1])
    elif args[0] == "--rounds":
        rounds = int(args[1])
    elif args[0] == "--coverage":
        coverage = int(args[1])
    args = args[2:]
res = {n: [] for n in args}
for r in range(rounds):
    for n in args:
        env = dict(os.environ)
        if n != "main":
            env["BSCALL_AMD_LIB"] = os.path.join(ROOT, "bs_call_amd", "lib", "variants", "lib_%s.so" % n)
        if chain:
            cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_chain.py"), "--no-unfused", "--steps", "7", "--sites", str(sites), "--coverage", str(coverage)]
        else:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-chain", "--steps", "5", "--warmup", "2",
                   "--sites", str(sites), "--coverage", str(coverage)]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[n].append(d["fused_device_ms_last_window"] if chain else d["roofline"]["kernel_ms_avg"])
        except Exception:
            print(n, "FAILED", out.stderr[-500:])
for n, v in res.items():
    if v:
        v = sorted(v)
        print("%-12s kernel_ms median %.3f min %.3f  (%d runs)  -> %.2f G sites/s" % (n, v[len(v) // 2], v[0], len(v), sites / v[len(v) // 2] / 1e6))
