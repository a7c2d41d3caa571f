cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sams_gpu.py -x -q -k "three_training_steps and progressive" > gpurun_out/r03_n_sams_prog.log 2>&1; grep -E "^E  " gpurun_out/r03_n_sams_prog.log | cut -c1-600 | head -5
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd" 2>&1 | tail -2
timeout 600 python tools/wino_bench.py 2>/dev/null | cut -d, -f1-8,16- > gpurun_out/r03_n_wino_bench.csv; cat gpurun_out/r03_n_wino_bench.csv
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table 2>/dev/null | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 2>/dev/null | cut -c1-200
