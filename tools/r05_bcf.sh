#!/bin/bash
# Round 5 (GPU box): the BCF encoder on the device — its tests, A/B of library variants (VARIANTS="main w2 ...") on tools/bench_bcf.py, a kernel
# trace of the main build, then the file-to-file bench (tools/bench_bam2bcf.py).  usage: bash tools/r05_bcf.sh <tag>   (SKIP_TESTS=1, SKIP_E2E=1)
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
[ -n "$SKIP_TESTS" ] || timeout -k 10 600 python3 -m pytest tests/test_gpu_bcf.py tests/test_gpu_pipeline.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -2 $O/pytest.txt
for rep in 1 2; do
for v in ${VARIANTS:-main}; do
  if [ $v != main ]; then export BSCALL_AMD_LIB=$ROOT/bs_call_amd/lib/variants/lib_$v.so; else unset BSCALL_AMD_LIB; fi
  timeout -k 10 200 python3 tools/bench_bcf.py > $O/bcf_$v.json 2> $O/bcf_$v.err || { tail -5 $O/bcf_$v.err; exit 1; }
  python3 -c "import json; r = json.load(open('$O/bcf_$v.json')); print('%-8s %.4f ms (min %.4f)  %.0f GB/s  %.2f G records/s' % ('$v', r['device_ms_avg'], r['device_ms_min'], r['achieved_GBps'], r['records_per_s'] / 1e9))"
done
done
unset BSCALL_AMD_LIB
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $ROOT/tools/bench_bcf.py --steps 6 > $O/prof_run.json 2> $O/prof_run.err ) || { tail -5 $O/prof_run.err; exit 1; }
python3 tools/kstats.py $O/prof > $O/kstats.txt 2>&1 || true; grep -i "bcf\|scan\|lookback" $O/kstats.txt | head -12 || true
[ -n "$SKIP_E2E" ] || { timeout -k 10 500 python3 tools/bench_bam2bcf.py ${E2E_SITES:-2000000} > $O/bam2bcf.json 2> $O/bam2bcf.err || { tail -5 $O/bam2bcf.err; exit 1; }; cat $O/bam2bcf.json; }
