# kernel trace of the configs[2] runner (rank 0 of 8): per-kernel time and the gaps between the windows' launches
set -e
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_cfg3 -- python3 $ROOT/bench.py --config 3 --rank-of 8 --steps 2 --warmup 1 "$@" > $ROOT/gpurun_out/cfg3_prof.json 2> $ROOT/gpurun_out/cfg3_prof.err || { tail -5 $ROOT/gpurun_out/cfg3_prof.err; exit 1; }
cd $ROOT
find gpurun_out/prof_cfg3 -name "*kernel_stats.csv" | xargs cat | cut -c1-200
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_cfg3/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [i for i, r in enumerate(rows) if "bsc_chain" in r["Kernel_Name"]]
# the last 200 chain-related launches: kernel, duration, gap to the previous kernel's end
prev_end = None
out = []
for r in rows[ch[0]:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:40], (e - s) / 1e3, None if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
import collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for n, d, g in out[-400:]:
    a = agg[n]; a[0] += 1; a[1] += d; a[2] += (g or 0.0)
for n, a in agg.items():
    print("%-42s n=%4d  avg_us=%9.2f  avg_gap_before_us=%8.2f" % (n, a[0], a[1] / a[0], a[2] / a[0]))
for n, d, g in out[-12:]:
    print(n, round(d, 2), g)
PY
