cd /root/repo
python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | grep -v "^$" | tail -40 | cut -c1-250
