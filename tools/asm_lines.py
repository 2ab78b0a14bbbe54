#!/usr/bin/env python3
"""Static instruction census of one kernel of a `hipcc -S -gline-tables-only` listing by source line: how many VALU / SALU /
LDS / VMEM instructions each source line (or each file) compiled into.  Loops are counted once (static).
usage: python tools/asm_lines.py <file.s> <kernel-symbol-substring> [--by-file] [--file NAME --bucket N]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
by_file = "--by-file" in sys.argv
only = sys.argv[sys.argv.index("--file") + 1] if "--file" in sys.argv else None
bucket = int(sys.argv[sys.argv.index("--bucket") + 1]) if "--bucket" in sys.argv else 1
files, cur, inside = {}, (0, 0), False
cnt = collections.defaultdict(lambda: collections.Counter())
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
        continue
    if not inside and re.match(r"[A-Za-z_][\w$.]*:", s) and sym in s.split(":")[0]:
        inside = True
        continue
    if inside and s.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith((".", ";", "//")) or re.match(r"[\w$.]+:", s):
        continue
    op = s.split()[0]
    kind = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop", "s_cbranch", "s_branch"))
            else "SMEM" if op.startswith(("s_load", "s_buffer")) else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_"))
            else "WAIT" if op.startswith(("s_waitcnt", "s_nop")) else "BR")
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        kind = "LANE"
    f = files.get(cur[0], "?")
    key = f if by_file else (f, cur[1] // bucket * bucket)
    cnt[key][kind] += 1
tot = collections.Counter()
for k in sorted(cnt, key=lambda k: (str(k))):
    c = cnt[k]
    tot.update(c)
    if only and (k if by_file else k[0]) != only:
        continue
    print("%-34s %s" % (k if by_file else "%s:%d" % k, "  ".join("%s %d" % kv for kv in sorted(c.items()))))
print("TOTAL", dict(tot))
