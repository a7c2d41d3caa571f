cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -k "bucketed or three_frames or golden" > gpurun_out/r03_g_tests_bucketed.log 2>&1; echo "bucketed tests rc=$?"
timeout 900 python -m pytest tests/test_00_multi_rank_gpu.py -x -q > gpurun_out/r03_g_tests_multirank.log 2>&1; echo "multirank rc=$?"
SHINEON_BUCKETED=1 timeout 300 python bench.py --no-cpu-baseline --config c3 > gpurun_out/r03_g_bench_c3_bucketed.json 2> gpurun_out/r03_g_bench_c3_bucketed.log; echo "bench c3 bucketed rc=$?"
timeout 300 python bench.py --no-cpu-baseline --config c3 > gpurun_out/r03_g_bench_c3.json 2> gpurun_out/r03_g_bench_c3.log; echo "bench c3 rc=$?"
tail -4 gpurun_out/r03_g_tests_bucketed.log; tail -4 gpurun_out/r03_g_tests_multirank.log; cut -c1-200 gpurun_out/r03_g_bench_c3_bucketed.json; cut -c1-200 gpurun_out/r03_g_bench_c3.json
