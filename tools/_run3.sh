cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd" > gpurun_out/r03_c_tests_wino.log 2>&1; echo "wino tests rc=$?"
timeout 600 python tools/wino_bench.py > gpurun_out/r03_c_wino_bench.csv 2> gpurun_out/r03_c_wino_bench.err; echo "wino bench rc=$?"
timeout 120 python tools/probes/graph_event_probe.py > gpurun_out/r03_c_graph_event_probe.txt 2>&1; echo "probe rc=$?"
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03_c_bench_c4.json 2> gpurun_out/r03_c_bench_c4.log; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --config c3 > gpurun_out/r03_c_bench_c3.json 2> gpurun_out/r03_c_bench_c3.log; echo "bench c3 rc=$?"
timeout 900 python -m pytest tests/test_parity_bs4_gpu.py tests/test_models_gpu.py -x -q > gpurun_out/r03_c_tests_parity.log 2>&1; echo "parity rc=$?"
timeout 600 python -m pytest tests/test_sams_gpu.py -x -q -s -k "full_size_generator" > gpurun_out/r03_c_tests_sams_gen.log 2>&1; echo "sams gen rc=$?"
tail -3 gpurun_out/r03_c_tests_wino.log; cat gpurun_out/r03_c_wino_bench.csv; cat gpurun_out/r03_c_graph_event_probe.txt; cut -c1-250 gpurun_out/r03_c_bench_c4.json; cut -c1-250 gpurun_out/r03_c_bench_c3.json; tail -5 gpurun_out/r03_c_tests_parity.log; tail -5 gpurun_out/r03_c_tests_sams_gen.log
