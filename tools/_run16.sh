cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_bs4_gpu.py -q -s > gpurun_out/r03_p_parity.log 2>&1
grep -E "^E  |gradient tensors compared|passed|failed" gpurun_out/r03_p_parity.log | cut -c1-700 | head -30
