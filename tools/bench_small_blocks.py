#!/usr/bin/env python3
"""Small blocks — the reference's real unit (one call_genotypes_ML per maximal run of overlapping templates,
src/get_template_vector.c:141-147: 10^2 .. 10^7 positions): host buffers in, packed records out, wall time per position.
  one by one   bsc_block_records per block (a dozen launches, four copies and one wait each)
  batched      bsc_blocks_records over batches of >= --batch positions (one launch sequence per batch)
  gt_vcf form  bsc_blocks_submit_to (staged, from ordinary memory) and bsc_blocks_submit_to_inplace (page-locked buffers) + bsc_block_fetch:
               208-byte images of every position, the form the drop-in glue runs (integration/amd_overlap_protocol.h)
  bytes form   bsc_block_bcf block by block against bsc_blocks_bcf_submit_inplace per batch (page-locked buffers): the BCF stream back
usage: python tools/bench_small_blocks.py [--batch 1000000] [--total 4000000] [--coverage 30] [--sizes 1000,10000,100000]"""
import argparse
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import bs_call_amd as B

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1_000_000)
ap.add_argument("--total", type=int, default=4_000_000)
ap.add_argument("--coverage", type=int, default=30)
ap.add_argument("--sizes", default="1000,10000,100000")
args = ap.parse_args()
res = {"coverage": args.coverage, "batch_positions": args.batch, "what": "wall time, host buffers in (pageable), packed records out; best of 5 passes", "sizes": {}}
with B.SiteCaller() as c:
    for n in [int(v) for v in args.sizes.split(",")]:
        k = max(2, args.total // n)
        blocks, refs, pos = [], [], 1000
        for i in range(k):
            tpl, seq = B.synth_reads_host(5 + i, pos + 200, n, args.coverage)
            x, y = pos + 198, int((tpl["pos"] + tpl["len"]).max()) - 1
            blocks.append((tpl, seq, x, y))
            refs.append(B.synth_ref_host(5 + i, x, y - x + 3))
            pos = y
        total = sum(y - x + 1 for _, _, x, y in blocks)
        out = np.zeros(total, dtype=B.VCF_REC)
        per_batch = max(1, args.batch // n)
        groups = [(blocks[i : i + per_batch], refs[i : i + per_batch]) for i in range(0, k, per_batch)]
        t_one, t_bat, n1, n2 = [], [], 0, 0
        for rep in range(6):
            t0 = time.perf_counter()
            n1 = sum(len(c.block_records(t, s, x, y, refs[i], out=out)) for i, (t, s, x, y) in enumerate(blocks))
            t1 = time.perf_counter()
            n2 = sum(len(c.blocks_records(g, r, out=out)[0]) for g, r in groups)
            t2 = time.perf_counter()
            if rep:  # the first pass sizes the workspaces
                t_one.append(t1 - t0)
                t_bat.append(t2 - t1)
        assert n1 == n2
        # what a C host does: it flattens its blocks straight into ONE set of page-locked buffers per batch (here: built once,
        # outside the timed passes) and submits them in place; records come back into a page-locked array
        from bs_call_amd.abi import BLOCK_DESC
        from bs_call_amd.caller import PinnedBuffer
        keep, joined = [], []
        for g, r in groups:
            nt, ns, nr_ = sum(len(b[0]) for b in g), sum(b[1].size for b in g), sum(len(v) for v in r)
            pt, ps, pr = PinnedBuffer(nt, B.TEMPLATE), PinnedBuffer(ns, np.uint8), PinnedBuffer(nr_, np.uint8)
            po = PinnedBuffer(sum(b[3] - b[2] + 1 for b in g), B.VCF_REC)
            desc = np.zeros(len(g), dtype=BLOCK_DESC)
            ot = os_ = orf = 0
            for i, (t, sq, x, y) in enumerate(g):
                pt.array[ot : ot + len(t)] = t
                pt.array["off"][ot : ot + len(t)] += np.uint64(os_)
                ps.array[os_ : os_ + sq.size] = sq
                pr.array[orf : orf + len(r[i])] = r[i]
                desc[i] = (x, y, len(t), 0)
                ot, os_, orf = ot + len(t), os_ + sq.size, orf + len(r[i])
            keep.append((pt, ps, pr, po))
            joined.append((desc, pt.array, ps.array, pr.array, po.array))
        t_pin, n3 = [], 0
        for rep in range(6):
            t0 = time.perf_counter()
            n3 = sum(len(c.blocks_records_joined(*j, inplace=True)[0]) for j in joined)
            if rep:
                t_pin.append(time.perf_counter() - t0)
        assert n3 == n1
        # ---- the gt_vcf form the drop-in glue runs (integration/amd_overlap_protocol.h): bsc_blocks_submit_to[_inplace] + bsc_block_fetch,
        # 208-byte images of every position back into a page-locked array; straight through the C ABI on arrays joined beforehand
        # (a) staged: inputs in ordinary memory, the library copies them to its pinned staging area (round 4's glue);
        # (b) in place: the batch built in page-locked buffers (round 5's glue)
        from bs_call_amd.caller import _ptr
        L, h = c._L, c._h
        P_max = max(sum(((b[3] - b[2] + 1 + 63) // 64) * 64 for b in g) for g, _ in groups)
        img, skp = PinnedBuffer((P_max, 208), np.uint8), PinnedBuffer(P_max, np.uint8)
        t_stage, t_inpl = [], []
        pageable = [(d, np.array(t), np.array(sq), np.array(rf)) for d, t, sq, rf, _ in joined]
        for rep in range(4):
            for fn, lst, acc in ((L.bsc_blocks_submit_to, pageable, t_stage), (L.bsc_blocks_submit_to_inplace, [j[:4] for j in joined], t_inpl)):
                t0 = time.perf_counter()
                for d, t, sq, rf in lst:
                    off = np.zeros(len(d), dtype=np.uint64)
                    rc = fn(h, _ptr(d), len(d), _ptr(t), _ptr(sq), sq.size, _ptr(rf), _ptr(img.array), 208, _ptr(skp.array), _ptr(off))
                    assert rc >= 0, rc
                    rc = L.bsc_block_fetch(h, None, None)
                    assert rc >= 0, rc
                if rep:
                    acc.append(time.perf_counter() - t0)
        img.free()
        skp.free()
        # ---- the bytes form: BCF stream back; one bsc_block_bcf_submit_inplace per block against
        # bsc_blocks_bcf_submit_inplace per batch, straight through the C ABI on the page-locked arrays joined above
        import ctypes as C

        from bs_call_amd import _lib
        ids = _lib.BcfIds()
        L.bsc_bcf_default_ids(C.byref(ids))
        vp = _lib.VcfParams(0, 1, 0xFFFFFFFF)
        cap_b = 128 * P_max + 4096
        bout = PinnedBuffer(cap_b, np.uint8)
        nb_, nr_c = C.c_uint64(0), C.c_uint64(0)
        t_b1, t_bb, bytes_1, bytes_b = [], [], 0, 0
        for rep in range(4):
            t0 = time.perf_counter()
            bytes_1 = sum(len(c.block_bcf(t, sq, x, y, refs[i], 0)[0]) for i, (t, sq, x, y) in enumerate(blocks))  # one by one, as the records above
            t1 = time.perf_counter()
            bytes_b = 0
            for d, t, sq, rf, _ in joined:
                rc = L.bsc_blocks_bcf_submit_inplace(h, _ptr(d), len(d), _ptr(t), _ptr(sq), sq.size, _ptr(rf), None, C.byref(vp), 0, 0, C.byref(ids), None, _ptr(bout.array),
                                                     cap_b)
                assert rc >= 0, rc
                rc = L.bsc_blocks_bcf_fetch(h, C.byref(nb_), C.byref(nr_c))
                assert rc >= 0, rc
                bytes_b += nb_.value
            t2 = time.perf_counter()
            if rep:
                t_b1.append(t1 - t0)
                t_bb.append(t2 - t1)
        assert bytes_1 == bytes_b
        bout.free()
        # the Python wrapper joins the blocks' arrays for the batched call (np.concatenate): timed apart, it is not the library's
        t0 = time.perf_counter()
        for g, r in groups:
            np.concatenate([b[0] for b in g]), np.concatenate([b[1] for b in g]), np.concatenate(r)
        join_s = time.perf_counter() - t0
        res["sizes"][str(n)] = {"blocks": k, "positions": total, "records": n1,
                                "one_by_one_M_positions_per_s": total / min(t_one) / 1e6, "one_by_one_us_per_block": min(t_one) / k * 1e6,
                                "batched_M_positions_per_s": total / min(t_bat) / 1e6, "batched_us_per_block": min(t_bat) / k * 1e6,
                                "batched_minus_python_join_M_positions_per_s": total / max(min(t_bat) - join_s, 1e-9) / 1e6,
                                "batched_inplace_pinned_M_positions_per_s": total / min(t_pin) / 1e6,
                                "batched_inplace_pinned_us_per_block": min(t_pin) / k * 1e6, "blocks_per_batch": per_batch,
                                "gt_vcf_images_staged_M_positions_per_s": total / min(t_stage) / 1e6,
                                "gt_vcf_images_inplace_pinned_M_positions_per_s": total / min(t_inpl) / 1e6,
                                "bcf_bytes_one_by_one_M_positions_per_s": total / min(t_b1) / 1e6,
                                "bcf_bytes_batched_inplace_M_positions_per_s": total / min(t_bb) / 1e6, "bcf_bytes": bytes_b}
        print("%7d-position blocks x %4d: one by one %8.1f M positions/s (%6.0f us per block)   batched %8.1f M positions/s (%6.0f us per block; "
              "%.1f without the wrapper's array joins)   batched, in place from page-locked buffers %8.1f M positions/s (%6.0f us per block)"
              % (n, k, total / min(t_one) / 1e6, min(t_one) / k * 1e6, total / min(t_bat) / 1e6, min(t_bat) / k * 1e6,
                 total / max(min(t_bat) - join_s, 1e-9) / 1e6, total / min(t_pin) / 1e6, min(t_pin) / k * 1e6), file=sys.stderr)
        print("%7s gt_vcf images (208 B per position back; what the drop-in glue runs): bsc_blocks_submit_to from ordinary memory %8.1f M positions/s"
              "   bsc_blocks_submit_to_inplace from page-locked buffers %8.1f M positions/s" % ("", total / min(t_stage) / 1e6, total / min(t_inpl) / 1e6),
              file=sys.stderr)
        print("%7s BCF bytes (57 B per position back; INTEGRATION.md 2b), in place from page-locked buffers: bsc_block_bcf block by block (ordinary memory) %8.1f M positions/s"
              "   bsc_blocks_bcf_submit_inplace per batch %8.1f M positions/s" % ("", total / min(t_b1) / 1e6, total / min(t_bb) / 1e6), file=sys.stderr)
        for pb in keep:
            for q in pb:
                q.free()
print(json.dumps(res))
