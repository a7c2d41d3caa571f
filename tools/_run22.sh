cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -q -m gpu --durations=6 > gpurun_out/r03_x_tests_full.log 2>&1; echo "full gpu tests rc=$?"
tail -12 gpurun_out/r03_x_tests_full.log
