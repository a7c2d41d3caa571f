cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -s -k "winograd" 2>&1 | grep -E "F\(4x4|passed|failed|Error" | head -12
timeout 600 python tools/wino_bench.py 2>/dev/null | cut -d, -f1-8,16- > gpurun_out/r03_o_wino_bench.csv; cat gpurun_out/r03_o_wino_bench.csv
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table 2>/dev/null | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 2>/dev/null | cut -c1-200
timeout 900 python -m pytest tests/test_parity_bs4_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -4
