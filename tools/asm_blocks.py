#!/usr/bin/env python3
"""Basic blocks of one kernel of a `hipcc -S -gline-tables-only` listing: per block its label, VALU / SALU / LDS / VMEM / lane-move
counts, the branch that ends it and the source lines it comes from (file:first-last) — with loop trip counts known from the data
this gives the dynamic instruction budget of a tile.  usage: python tools/asm_blocks.py <file.s> <kernel-symbol-substring> [min_valu]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 0
files, inside, cur = {}, False, None
blocks = []


def new_block(label):
    global cur
    cur = {"label": label, "c": collections.Counter(), "lines": collections.defaultdict(lambda: [10**9, 0, 0]), "end": ""}
    blocks.append(cur)


loc = (0, 0)
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
        continue
    if not inside and re.match(r"[A-Za-z_][\w$.]*:", s) and sym in s.split(":")[0]:
        inside = True
        new_block("entry")
        continue
    if inside and s.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        loc = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"(\.LBB[\w$.]+):", s)
    if m:
        new_block(m.group(1))
        continue
    if not s or s.startswith((".", ";", "//")):
        continue
    op = s.split()[0]
    kind = ("LANE" if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) else "VALU" if op.startswith("v_") else
            "BR" if op.startswith(("s_cbranch", "s_branch")) else "SMEM" if op.startswith(("s_load", "s_buffer")) else
            "WAIT" if op.startswith(("s_waitcnt", "s_nop")) else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else
            "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "OTHER")
    cur["c"][kind] += 1
    f = files.get(loc[0], "?")
    r = cur["lines"][f]
    r[0], r[1], r[2] = min(r[0], loc[1]), max(r[1], loc[1]), r[2] + (1 if kind == "VALU" else 0)
    if kind == "BR":
        cur["end"] = s
tot = collections.Counter()
for b in blocks:
    tot.update(b["c"])
    if b["c"]["VALU"] < min_valu:
        continue
    src = "  ".join("%s:%d-%d(%d)" % (f, r[0], r[1], r[2]) for f, r in sorted(b["lines"].items(), key=lambda kv: -kv[1][2])[:4])
    print("%-12s V %4d L %3d S %4d LDS %3d VM %2d | %-28s | %s" % (b["label"], b["c"]["VALU"], b["c"]["LANE"], b["c"]["SALU"], b["c"]["LDS"],
                                                              b["c"]["VMEM"], b["end"][:28], src))
print("TOTAL", dict(tot))
