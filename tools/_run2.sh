cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -s -k "winograd" > gpurun_out/r03_b_tests_wino.log 2>&1; echo "wino tests rc=$?"
timeout 600 python tools/wino_bench.py > gpurun_out/r03_b_wino_bench.csv 2> gpurun_out/r03_b_wino_bench.err; echo "wino bench rc=$?"
timeout 600 python -m pytest tests/test_parity_bs4_gpu.py -x -q -s -k "chained" > gpurun_out/r03_b_tests_parity.log 2>&1; echo "parity rc=$?"
timeout 400 python -m pytest tests/test_sams_gpu.py -x -q -k "three_training_steps and attn_gelu" > gpurun_out/r03_b_tests_sams3.log 2>&1; echo "sams3 rc=$?"
timeout 900 python -m pytest tests/test_sams_gpu.py -x -q -s -k "full_size_three or full_size_generator" --durations=5 > gpurun_out/r03_b_tests_sams_full.log 2>&1; echo "sams_full rc=$?"
tail -15 gpurun_out/r03_b_tests_wino.log; cat gpurun_out/r03_b_wino_bench.csv; tail -3 gpurun_out/r03_b_tests_parity.log; tail -3 gpurun_out/r03_b_tests_sams3.log; tail -12 gpurun_out/r03_b_tests_sams_full.log
