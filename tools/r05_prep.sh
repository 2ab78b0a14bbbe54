#!/bin/bash
# Device pre-processing: its tests, then tools/bench_prep.py under the kernel trace (GPU box).  usage: bash tools/r05_prep.sh <tag>
TAG=${1:-r05_prep}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_prep.py tests/test_gpu_pipeline.py tests/test_gpu_api_errors.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $ROOT/tools/bench_prep.py > $O/prep.json 2> $O/err.txt) || { tail $O/err.txt; exit 1; }
python3 tools/kstats_timed.py $O/t 2 bsc_ > $O/kern.txt
cat $O/kern.txt $O/prep.json
