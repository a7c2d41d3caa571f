cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_sams_gpu.py -x -q -k "three_training_steps or full_size_generator" 2>&1 | tail -3
bash tools/profile_round.sh r03_m > gpurun_out/r03_m_profile_round.log 2>&1
tail -40 gpurun_out/r03_m_profile_round.log | cut -c1-300
