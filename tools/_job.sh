cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_models_gpu.py tests/test_parity_bs4_gpu.py -x -q -m gpu 2>&1 | tail -4
for c in c3 c3 c4 c4; do python3 bench.py --config $c --no-cpu-baseline --no-hbm-table --steps 40 2>/dev/null | grep "^{" | cut -c1-150; done
