set -e
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q > gpurun_out/r2_pytest5.log 2>&1 || { tail -30 gpurun_out/r2_pytest5.log; exit 1; }
tail -3 gpurun_out/r2_pytest5.log
bash tools/profile_bench.sh r02_a --steps 10 --warmup 3 > gpurun_out/prof_r02_a.log 2>&1 || { tail -20 gpurun_out/prof_r02_a.log; exit 1; }
tail -12 gpurun_out/prof_r02_a.log
python tools/make_traffic_json.py gpurun_out/prof_r02_a
bash tools/pmc_sq.sh r02_a > gpurun_out/sq_r02_a.txt 2>&1 || { tail -20 gpurun_out/sq_r02_a.txt; exit 1; }
cat gpurun_out/sq_r02_a.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r02_chain -- python3 $ROOT/tools/bench_chain.py --steps 8 > $ROOT/gpurun_out/r2_chain_prof.json 2> $ROOT/gpurun_out/r2_chain_prof.err || { tail -5 $ROOT/gpurun_out/r2_chain_prof.err; exit 1; }
cd $ROOT
cat gpurun_out/r2_chain_prof.json
find gpurun_out/prof_r02_chain -name "*kernel_stats.csv" | xargs cat | cut -c1-160
