cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd" 2>&1 | tail -2
timeout 600 python tools/wino_bench.py --layers 0 1 2 3 4 2>/dev/null | cut -d, -f1-8,16- > gpurun_out/r03_y_wino_bench.csv; cat gpurun_out/r03_y_wino_bench.csv
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table 2>/dev/null | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 2>/dev/null | cut -c1-200
