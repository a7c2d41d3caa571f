cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -s -k "winograd" 2>&1 | grep -E "F\(4x4|winograd |passed|failed|Error" | head -14
timeout 900 python -m pytest tests/test_parity_bs4_gpu.py -q -s > gpurun_out/r03_q_parity.log 2>&1
grep -E "^E  |gradient tensors compared|passed|failed" gpurun_out/r03_q_parity.log | cut -c1-500 | head -30
timeout 600 python -m pytest tests/test_models_gpu.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table 2>/dev/null | cut -c1-200
