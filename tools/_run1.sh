cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
free -g | head -2 > gpurun_out/r03_a_host.txt; nproc >> gpurun_out/r03_a_host.txt
timeout 1500 python -m pytest tests/test_sams_gpu.py -x -q -s -k "full_size" --durations=10 > gpurun_out/r03_a_tests_sams_full.log 2>&1; echo "sams_full rc=$?" 
timeout 900 python -m pytest tests/test_parity_bs4_gpu.py tests/test_models_gpu.py -x -q -s --durations=10 > gpurun_out/r03_a_tests_parity.log 2>&1; echo "parity rc=$?"
timeout 600 python -m pytest tests/test_sams_gpu.py -x -q -k "three_training_steps" > gpurun_out/r03_a_tests_sams3.log 2>&1; echo "sams3 rc=$?"
timeout 300 python tools/dgamma_probe.py > gpurun_out/r03_a_dgamma.log 2>&1; echo "dgamma rc=$?"
timeout 300 python bench.py > gpurun_out/r03_a_bench_c4.json 2> gpurun_out/r03_a_bench_c4.log; echo "bench rc=$?"
tail -3 gpurun_out/r03_a_tests_sams_full.log; tail -3 gpurun_out/r03_a_tests_parity.log; tail -3 gpurun_out/r03_a_tests_sams3.log; tail -12 gpurun_out/r03_a_dgamma.log; cut -c1-300 gpurun_out/r03_a_bench_c4.json
