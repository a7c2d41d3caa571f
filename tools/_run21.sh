cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_sams_gpu.py -x -q -s -k "full_size_generator" > gpurun_out/r03_v_sams_gen.log 2>&1
grep -E "^E  |sams generator" gpurun_out/r03_v_sams_gen.log | cut -c1-700 | head
