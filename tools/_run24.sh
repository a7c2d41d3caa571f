cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd or conv" 2>&1 | tail -2
timeout 300 python bench.py --no-hbm-table 2>/dev/null | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
