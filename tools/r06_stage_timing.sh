set -e
T=/tmp/st; mkdir -p $T; cd $GRAFT_REPO_ROOT
gcc -O2 -o $T/gen tools/make_wgbs_bam.c -lz -lpthread -lm
$T/gen $T/in.bam $T/ref.fa 4000000 2 7 8 1 0 1 1 > /dev/null
BSC_STAGE_TIMING=1 BAM2BCF_TIMING=1 bs_call_amd/lib/bam2bcf $T/in.bam $T/ref.fa $T/out.bcf $T/rep.json 2> $T/err.txt > $T/out.txt || true
grep "bsc stage" $T/err.txt | sed 's/ [0-9.]* us$//' | sort | uniq -c | sort -rn | head
python3 - <<'PY'
import re,collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for l in open('/tmp/st/err.txt'):
    m=re.match(r"bsc stage: (.*) ([0-9.]+) us$", l.strip())
    if m: acc[m.group(1)]+=float(m.group(2))*1e-6; cnt[m.group(1)]+=1
for k,v in sorted(acc.items(), key=lambda kv:-kv[1]): print("%-70s total %.3f s  n %d  avg %.1f us" % (k, v, cnt[k], v/cnt[k]*1e6))
PY
tail -2 $T/err.txt | cut -c1-400
