cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 tools/probes/stream_wait_value.bin > gpurun_out/r03_d_stream_wait_value.txt 2>&1; echo "probe rc=$?"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_parity_bs4_gpu.py -x -q > gpurun_out/r03_d_tests_main.log 2>&1; echo "main tests rc=$?"
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03_d_bench_c4.json 2> gpurun_out/r03_d_bench_c4.log; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --config c3 > gpurun_out/r03_d_bench_c3.json 2> gpurun_out/r03_d_bench_c3.log; echo "bench c3 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --config c2 > gpurun_out/r03_d_bench_c2.json 2> gpurun_out/r03_d_bench_c2.log; echo "bench c2 rc=$?"
timeout 600 python bench.py --no-cpu-baseline --config sams --no-hbm-table > gpurun_out/r03_d_bench_sams.json 2> gpurun_out/r03_d_bench_sams.log; echo "bench sams rc=$?"
timeout 900 python -m pytest tests/test_sams_gpu.py -x -q -k "not full_size" > gpurun_out/r03_d_tests_sams.log 2>&1; echo "sams tests rc=$?"
cat gpurun_out/r03_d_stream_wait_value.txt; tail -5 gpurun_out/r03_d_tests_main.log; for c in c4 c3 c2 sams; do cut -c1-260 gpurun_out/r03_d_bench_$c.json; done; tail -5 gpurun_out/r03_d_tests_sams.log
