cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for mode in 1 0; do
  SHINEON_WAIT_MODE=$mode SHINEON_BUCKETED=1 SHINEON_BUCKET_MB=16 timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 2> gpurun_out/r03_i_c3_mode$mode.log | cut -c1-200 | sed "s/^/wait_mode=$mode: /"
done
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 2>/dev/null | cut -c1-200 | sed "s/^/plain: /"
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -k "bucketed or golden" 2>&1 | tail -3
