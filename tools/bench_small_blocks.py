import sys, time
sys.path.insert(0, '.')
import numpy as np
import bs_call_amd as B
with B.SiteCaller() as c:
    for n in (1_000, 10_000, 100_000, 1_000_000):
        tpl, seq = B.synth_reads_host(5, 1000, n, 30)
        x, y = 998, int((tpl["pos"] + tpl["len"]).max()) - 1
        nn = y - x + 1
        ref = B.synth_ref_host(5, x, nn + 2)
        out = np.zeros(nn, dtype=B.GT_METH); skip = np.zeros(nn, dtype=np.uint8); rec = np.zeros(nn, dtype=B.VCF_REC)
        c.call_block(tpl, seq, x, y, ref[:nn], out=out, skip=skip); c.block_records(tpl, seq, x, y, ref, out=rec)
        ts = []; tr = []
        for _ in range(20):
            t0 = time.perf_counter(); c.call_block(tpl, seq, x, y, ref[:nn], out=out, skip=skip); ts.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); c.block_records(tpl, seq, x, y, ref, out=rec); tr.append(time.perf_counter() - t0)
        print("%8d positions: call_block %.0f us (%.1f M/s), block_records %.0f us (%.1f M/s)" % (nn, min(ts) * 1e6, nn / min(ts) / 1e6, min(tr) * 1e6, nn / min(tr) / 1e6))
