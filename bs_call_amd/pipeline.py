"""BAM -> BCF through the library, block by block — the reference's four threads in a line (reader: bsc_bam_next_block;
process: the block's reference; process + calc + print: bsc_block_bcf_raw on the GPU — pre-processing, calling, record formation and
the BCF encoding; output: the writer) —
and the run's JSON report.  An example of the pieces put together for tests and for INTEGRATION.md, not a command-line
replacement of bs_call: the reference's argument parsing, region / contig selection and FASTA reader are out of scope
(SURVEY.md section 8)."""
from typing import Dict, Optional

import numpy as np

from . import report, vcf
from .bam import BamReader, block_reference
from .caller import ReadProfile, SiteCaller, gc_bins, prepare_templates


def run(bam_path: str, reference: Dict[str, np.ndarray], bcf_path: str, sample: str = "SAMPLE", report_path: Optional[str] = None,
        caller: Optional[SiteCaller] = None, dbsnp=None, compressed: bool = True, date=None, left_trim=(0, 0), right_trim=(0, 0),
        min_qual: Optional[int] = None, benchmark_mode: bool = False, under_conv: Optional[float] = None, over_conv: Optional[float] = None,
        host_prep: bool = False, host_bcf: bool = False, device_reader: bool = False, shard_rank: Optional[int] = None, shard_world: int = 1,
        reduce_device=None, **reader_kw) -> dict:
    """reference: contig name -> uint8 reference codes (0 = N, 1..4 = ACGT; position 1 first).  Returns a summary dict.
    under_conv / over_conv / min_qual (defaults 0.01 / 0.05 / 20, src/init_param.c:26-31) are the MODEL's parameters: without
    `caller` the run builds its SiteCaller from them; with one, they are taken from it and a differing explicit value is an error
    (the header must name the thresholds the genotypes were computed with, src/print_vcf.c:647-692).  host_prep: the read
    pre-processing on the host (bsc_prepare_templates_profile) instead of the device (round 5's default) — same bytes.  host_bcf:
    the packed records come back and the host encodes them (bsc_bcf_block) instead of the device's encoder (bsc_block_bcf_raw,
    round 5's default) — same bytes.  device_reader: the blocks are formed on the device from the inflated BAM bytes (round 6:
    bamdev.DeviceBamReader — the host only inflates) and go on to bsc_block_bcf_rawdev where they lie — same bytes.
    shard_rank / shard_world: ONE RANK of a sharded run over one file (SURVEY.md 8e on real input; the reference's unit of parallelism is a
    process per contig over an indexed file, src/process.c:125, src/get_template_vector.c:69-99): the header's contigs are dealt to the ranks by
    shard.assign_contigs (longest first), every rank reads the stretches of the file that hold ITS contigs (the device reader's contig
    selection: no index file), writes each contig's records to a shard beside bcf_path, and at the end the ranks all-reduce what the report
    sums (torch.distributed's default group: RCCL between GPUs, gloo in the rehearsal; reduce_device = the tensors' device) and rank 0
    concatenates the shards in contig order behind the header — the bytes of the single run."""
    own = caller is None
    if own:
        under_conv = 0.01 if under_conv is None else under_conv
        over_conv = 0.05 if over_conv is None else over_conv
        min_qual = 20 if min_qual is None else min_qual
        c = SiteCaller(under_conv=under_conv, over_conv=over_conv, min_qual=min_qual)
    else:
        c = caller
        for name, given in (("under_conv", under_conv), ("over_conv", over_conv), ("min_qual", min_qual)):
            if given is not None and given != c.params[name]:
                raise ValueError("%s=%r disagrees with the supplied caller's %r" % (name, given, c.params[name]))
        under_conv, over_conv, min_qual = c.params["under_conv"], c.params["over_conv"], c.params["min_qual"]
    try:
        prof = ReadProfile()
        base_filter = np.zeros(5, dtype=np.uint64)
        passed = np.zeros(2, dtype=np.uint64)
        per_contig = []
        n_blocks = n_records = 0
        c.reset_site_stats()
        sharded = shard_rank is not None
        if sharded:
            from . import shard as S
            from .bamdev import BamStream

            device_reader = True
            with BamStream(bam_path, threads=1, contigs=[]) as hs:  # the header alone
                all_refs = hs.refs
            parts = S.assign_contigs([l for _, l in all_refs], shard_world)
            mine = sorted(parts[shard_rank])
            if all_refs and len(all_refs) - 1 in parts[shard_rank]:
                mine = mine + [-1]  # the unplaced reads at the file's end go with the last contig's rank
            reader_kw = dict(reader_kw, contigs=mine)
            shard_files = {}
        if device_reader:
            if host_prep or host_bcf:
                raise ValueError("device_reader goes with the device pre-processing and encoder")
            from .bamdev import DeviceBamReader

            reader = DeviceBamReader(c, bam_path, **reader_kw)
        else:
            reader = BamReader(bam_path, **reader_kw)
        with reader as rd:
            refs = rd.refs
            # the header names the run's own thresholds (print_vcf_header, src/print_vcf.c:647-692)
            header = vcf.header_text([(n, l) for n, l in refs], sample, under_conv=under_conv, over_conv=over_conv,
                                     mapq_thresh=reader_kw.get("mapq_thresh", 20), min_qual=min_qual, date=date,
                                     dbsnp_header=None if dbsnp is None else dbsnp.header, benchmark_mode=benchmark_mode)

            def block_blobs():
                nonlocal n_blocks, n_records, base_filter, passed
                before, cur_tid = c.site_totals(), -1
                for item in (rd.device_blocks() if device_reader else rd.blocks()):
                    if device_reader:
                        dblk = item
                        tid, y = int(dblk.tid), int(dblk.y)
                    else:
                        tid, y, raw, seq, ms = item
                    name, _ = refs[tid]
                    if tid != cur_tid:
                        if cur_tid >= 0:
                            after = c.site_totals()
                            per_contig.append((refs[cur_tid][0], after - before))
                            before = after
                        cur_tid = tid
                        if dbsnp is not None:
                            dbsnp.load_contig(name)
                        # the contig's GC bins (load_sequence computes them when a report is asked for), resident on the device
                        gc_start, bins = gc_bins(reference[name])
                        c.set_gc_bins_host(bins, gc_start)
                    codes = reference[name]
                    if device_reader:
                        x = int(dblk.x)
                    else:
                        x = int(raw["pos"][0][0]) or int(raw["pos"][0][1])
                        x = x - 2 if x > 2 else 1  # process_template_vector, src/process_template.c:22-28
                    ref = block_reference(codes, x, y)
                    flags = None if dbsnp is None else dbsnp.flags(x, y - x + 1)
                    if device_reader:  # the block is in HBM already: pre-processing, calling, encoding behind it
                        names = None if dbsnp is None else dbsnp.names(x, y - x + 1)
                        blob, n_rec, st = c.block_bcf_rawdev(dblk, ref, tid, names=names, left_trim=left_trim, right_trim=right_trim, min_qual=min_qual,
                                                             reg_stop=len(codes), dbsnp=flags, with_stats=True, profile=prof)
                        recs = None
                    elif host_prep:  # round 4's split: the process thread's per-template work here, then the block
                        tpl, pseq, st = prepare_templates(raw, seq, ms, left_trim, right_trim, min_qual, profile=prof, x=x, ref=ref)
                        recs = c.block_records(tpl, pseq, x, y, ref, reg_stop=len(codes), dbsnp=flags, with_stats=True)
                    elif host_bcf:  # raw templates up, pre-processing and the read profile on the device (bsc_block_records_raw)
                        recs, st = c.block_records_raw(raw, seq, ms, x, y, ref, left_trim, right_trim, min_qual, reg_stop=len(codes), dbsnp=flags,
                                                       with_stats=True, profile=prof)
                    else:  # ... and the BCF encoding too: the block's stream comes back (bsc_block_bcf_raw)
                        names = None if dbsnp is None else dbsnp.names(x, y - x + 1)
                        blob, n_rec, st = c.block_bcf_raw(raw, seq, ms, x, y, ref, tid, names=names, left_trim=left_trim, right_trim=right_trim,
                                                          min_qual=min_qual, reg_stop=len(codes), dbsnp=flags, with_stats=True, profile=prof)
                        recs = None
                    base_filter += np.array([st["base_none"], st["base_trim"], st["base_clip"], st["base_overlap"], st["base_lowqual"]], dtype=np.uint64)
                    passed += np.array([st["reads"], st["read_bases"]], dtype=np.uint64)
                    if recs is not None:
                        blob, n_rec = vcf.bcf_block(recs, tid, dbsnp), len(recs)
                    if sharded:  # this contig's shard, written as its blocks are formed
                        if tid not in shard_files:
                            shard_files[tid] = open("%s.shard%05d" % (bcf_path, tid), "wb")
                        shard_files[tid].write(blob)
                    else:
                        yield blob
                    n_blocks += 1
                    n_records += n_rec
                if cur_tid >= 0:
                    per_contig.append((refs[cur_tid][0], c.site_totals() - before))

            if sharded:
                for _ in block_blobs():
                    pass
                for f_ in shard_files.values():
                    f_.close()
            else:
                vcf.write_bcf(bcf_path, header, block_blobs(), compressed)  # blocks go to the writer as they are formed
            cts, bases = rd.filter_counts()
        cts[0] += int(passed[0])
        bases[0] += int(passed[1])
        c.set_gc_bins_host(None, 0)
        site_stats, gc = c.site_stats(), c.gc_stats()
        prof_rows = prof.reported()
        if sharded:  # everything the report holds is a sum over positions / reads / templates: the ranks' shares add
            import torch.distributed as dist

            site_stats = S.allreduce_site_stats(site_stats, reduce_device)
            gc = S.allreduce_counts(gc, reduce_device)
            small = np.zeros(15 + 15 + 5 + 2 + 1, dtype=np.uint64)
            small[:15], small[15:30], small[30:35] = cts, bases, base_filter
            small[35], small[36], small[37] = n_blocks, n_records, 0
            small = S.allreduce_counts(small, reduce_device)
            cts, bases, base_filter = [int(v) for v in small[:15]], [int(v) for v in small[15:30]], small[30:35]
            n_blocks, n_records = int(small[35]), int(small[36])
            # the read profile: the vectors add; its length is the longest any rank saw
            used = np.array([prof.used], dtype=np.uint64)
            rows = S.allreduce_counts(prof.counts, reduce_device)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                import torch

                t = torch.from_numpy(used.view(np.int64).copy())
                if reduce_device is not None:
                    t = t.to(reduce_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                used = t.cpu().numpy().view(np.uint64)
            prof_rows = rows[: int(used[0])]
            # per contig: each is one rank's; a sum of the zero-padded table is a gather
            tab = np.zeros((len(refs), 14), dtype=np.uint64)
            names = [n for n, _ in refs]
            for n_, v in per_contig:
                tab[names.index(n_)] += np.asarray(v, dtype=np.uint64).reshape(-1)
            tab = S.allreduce_counts(tab, reduce_device)
            seen = np.zeros(len(refs), dtype=np.uint64)
            for n_, _v in per_contig:
                seen[names.index(n_)] = 1
            seen = S.allreduce_counts(seen, reduce_device)
            per_contig = [(names[i], tab[i].reshape(7, 2)) for i in range(len(refs)) if seen[i]]
            if dist.is_available() and dist.is_initialized():
                dist.barrier()  # every shard is on disk
        text = report.render_json(site_stats, gc=gc, min_qual=min_qual, date=date, have_dbsnp=dbsnp is not None, filter_cts=cts, filter_bases=bases,
                                  base_filter=np.asarray(base_filter).tolist(), read_profile=prof_rows, contigs=per_contig)
        if sharded and shard_rank == 0:  # the header, then the contigs' shards in the header's order
            import os

            def shards():
                for tid in range(len(refs)):
                    p_ = "%s.shard%05d" % (bcf_path, tid)
                    if os.path.exists(p_):
                        with open(p_, "rb") as f_:
                            while True:
                                b_ = f_.read(1 << 24)
                                if not b_:
                                    break
                                yield b_
                        os.remove(p_)

            vcf.write_bcf(bcf_path, header, shards(), compressed)
        if report_path and (not sharded or shard_rank == 0):
            with open(report_path, "w") as f:
                f.write(text)
        return {"blocks": n_blocks, "records": n_records, "report": text, "filter_cts": cts, "contigs": [n for n, _ in per_contig]}
    finally:
        if own:
            c.close()
