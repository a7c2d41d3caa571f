#!/bin/bash
# On a GPU box (via gpurun, from the repo root): ADDS plans to the committed file - only layer shapes that have none are
# measured (thorough mode); every existing plan, hence every existing bit pattern, stays.  Shapes come from the GPU tests and
# from bench.py --config c1.  Result: gpurun_out/gfx950.txt -> copy to shineon-virtual-tryon_amd/plans/gfx950.txt and commit.
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/gfx950.txt
cp shineon-virtual-tryon_amd/plans/gfx950.txt $OUT
wc -l $OUT
export SHINEON_AUTOTUNE=2
for t in tests/test_parity_bs4_gpu.py tests/test_sams_gpu.py tests/test_models_gpu.py tests/test_ops_gpu.py; do
  SHINEON_PLANS=$OUT SHINEON_PLANS_SAVE=$OUT SHINEON_ROUTES_NOCHECK=1 python -m pytest $t -q -m gpu -p no:cacheprovider 2>&1 | tail -2
  wc -l $OUT
done
SHINEON_PLANS=$OUT python3 bench.py --config c1 --plans $OUT --no-cpu-baseline --no-hbm-table 2>&1 | grep "plans\|^{" | cut -c1-160
wc -l $OUT
