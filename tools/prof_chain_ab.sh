# kernel trace of the 50 M-position chain bench for the main library and one variant, same box
set -e
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for V in main "$@"; do
  cd /tmp
  if [ "$V" = main ]; then unset BSCALL_AMD_LIB; else export BSCALL_AMD_LIB=$ROOT/bs_call_amd/lib/variants/lib_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_ab_$V -- python3 $ROOT/bench.py --config 3 --rank-of 8 --steps 2 --warmup 1 > $ROOT/gpurun_out/prof_ab_$V.json 2> $ROOT/gpurun_out/prof_ab_$V.err || { tail -5 $ROOT/gpurun_out/prof_ab_$V.err; exit 1; }
  cd $ROOT
  echo "== $V"
  find gpurun_out/prof_ab_$V -name "*kernel_stats.csv" | xargs cat | grep -i "chain" | cut -c1-60,330-460
  python3 -c "
import json; d=json.loads(open('gpurun_out/prof_ab_$V.json').read().strip().splitlines()[-1]); print(d['value']/1e9, d['ms_per_step'])"
done
