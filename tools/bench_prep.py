#!/usr/bin/env python3
"""Read pre-processing on the device (bsc_prepare_templates_device, csrc/prepdev.hip) at config size: the L-reads of a block handed
over RAW (the span is the length; every --indel-every-th read gets a 2-base deletion from the reference and a 1-base insertion,
so that the list logic and the padded copy run), device-resident, wall time of the call (it waits for the prepared size).  The
first chunk's templates are checked against the host form (csrc/prep.c).  Prints one JSON line; run under `rocprofv3
--kernel-trace --stats` for the two kernels' times.
usage: python tools/bench_prep.py [--sites N] [--coverage C] [--steps K] [--indel-every 50]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bs_call_amd as B
from bs_call_amd import reads as R
from bs_call_amd.abi import MISMS, PREP_PARAMS, PREP_STATS, RAW_TEMPLATE, TEMPLATE
from bs_call_amd.caller import _ptr, prepare_templates

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=50_000_000)
ap.add_argument("--coverage", type=int, default=30)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--indel-every", type=int, default=50)
ap.add_argument("--profile", action="store_true", help="with the non-CpG read profile (the form the pipeline runs)")
ap.add_argument("--overlap-every", type=int, default=0, help="every k-th pair's mate moved to 60 bases behind read 0: overlapping mates (handle_overlap runs)")
args = ap.parse_args()
dev = torch.device("cuda:0")
tpl, seq, y = R.synth_block(88172645463325252 + 2, 1000, args.sites, args.coverage)
raw = np.zeros(len(tpl), dtype=RAW_TEMPLATE)
for f in ("pos", "len", "off", "mapq", "orientation", "bs_strand"):
    raw[f] = tpl[f]
raw["reference_span"] = tpl["len"]
if args.overlap_every:  # short-insert pairs, as most WGBS libraries have them: the mate starts inside read 0
    ov = np.nonzero((np.arange(len(raw)) % args.overlap_every == 0) & (raw["len"][:, 0] >= 80) & (raw["len"][:, 1] > 0))[0]
    raw["pos"][ov, 1] = raw["pos"][ov, 0] + 60
# every k-th read 0 of at least 60 bases: a deletion from the reference at 20 (INS, 2 bases) and an insertion at 40 (DEL, 1 base)
sel = np.nonzero((np.arange(len(raw)) % args.indel_every == 0) & (raw["len"][:, 0] >= 60))[0]
ms = np.zeros(2 * len(sel), dtype=MISMS)
ms["type"][0::2], ms["position"][0::2], ms["size"][0::2] = 1, 20, 2
ms["type"][1::2], ms["position"][1::2], ms["size"][1::2] = 2, 40, 1
raw["n_misms"][sel, 0] = 2
raw["misms_off"][sel, 0] = 2 * np.arange(len(sel))
raw["reference_span"][sel, 0] += 1  # + 2 - 1
par = np.zeros(1, dtype=PREP_PARAMS)
par["min_qual"] = 20
cap = int(seq.size) + 2 * len(sel) + 16
res = {"positions": args.sites, "coverage": args.coverage, "templates": int(len(raw)), "bases": int(seq.size), "list_entries": int(len(ms))}
with B.SiteCaller() as c:
    up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
    d_raw, d_seq, d_ms = up(raw), up(seq), up(ms)
    d_tpl = torch.empty(len(raw) * TEMPLATE.itemsize, dtype=torch.uint8, device=dev)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    used, st = C.c_uint64(0), np.zeros(1, dtype=PREP_STATS)
    pf = None
    if args.profile:  # the block's reference codes (x .. y + 2) on the device, the counts on the host
        from bs_call_amd import _lib
        from bs_call_amd.caller import ReadProfile

        x = max(1, int(min(int(raw["pos"][0, 0]) or 1 << 30, int(raw["pos"][0, 1]) or 1 << 30)) - 2)
        y = int((raw["pos"].astype(np.int64) + raw["reference_span"]).max()) + 2
        d_ref = up(B.synth_ref_host(7, x, y - x + 3))
        prof = ReadProfile(cap=4096)
        pf = _lib.ReadProfile(d_ref.data_ptr(), x, y - x + 3, prof.counts.ctypes.data, prof.counts.shape[0], 0)
    wall = []
    for it in range(2 + args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = c._L.bsc_prepare_templates_device(c._h, d_raw.data_ptr(), len(raw), d_seq.data_ptr(), seq.size, d_ms.data_ptr(), len(ms), _ptr(par),
                                               d_tpl.data_ptr(), d_out.data_ptr(), cap, C.byref(used), _ptr(st), None if pf is None else C.byref(pf), None)
        assert rc == 0, c._L.bsc_last_error()
        if it >= 2:
            wall.append(time.perf_counter() - t0)
    w = float(np.median(wall))
    res["read_profile"] = bool(args.profile)
    res["calls"] = 2 + args.steps  # of bsc_prepare_templates_device in this process (tools/make_r06_json.py divides the PMC sums by it)
    res.update(wall_ms=w * 1e3, templates_per_s=len(raw) / w, bases_per_s=seq.size / w, positions_per_s=args.sites / w,
               bytes_in_plus_out=int(seq.size + used.value + len(raw) * (72 + 40) + len(ms) * 12),
               GBps=(seq.size + used.value + len(raw) * 112 + len(ms) * 12) / w / 1e9, prepared_bytes=int(used.value))
    # the first 200 000 templates against the host form
    m = min(200_000, len(raw))
    h_tpl, h_seq, h_st = prepare_templates(raw[:m], seq, ms)
    g_tpl = d_tpl[: m * 40].cpu().numpy().view(TEMPLATE)
    g_seq = d_out[: int(h_seq.size)].cpu().numpy()
    res["first_templates_equal_host_form"] = bool(g_tpl.tobytes() == h_tpl.tobytes() and g_seq.tobytes() == h_seq.tobytes())
print(json.dumps(res))
