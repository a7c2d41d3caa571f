cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 > gpurun_out/r03_l_tests_full.log 2>&1; echo "full gpu tests rc=$?"
tail -30 gpurun_out/r03_l_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
