#!/bin/bash
# On a GPU box (via gpurun, from the repo root): regenerates the committed igemm plans in thorough mode and appends the plans of
# every further layer shape the GPU tests launch (SAMS small / full-size cases, graph-replayed C5, un-graphed models).
# Result: gpurun_out/gfx950.txt -> copy to shineon-virtual-tryon_amd/plans/gfx950.txt and commit.
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/gfx950.txt
export SHINEON_AUTOTUNE=2
SHINEON_PLANS=none python tools/make_plans.py $OUT > gpurun_out/make_plans.log 2>&1
tail -3 gpurun_out/make_plans.log
for t in tests/test_parity_bs4_gpu.py tests/test_sams_gpu.py tests/test_models_gpu.py tests/test_ops_gpu.py; do
  SHINEON_PLANS=$OUT SHINEON_PLANS_SAVE=$OUT SHINEON_ROUTES_NOCHECK=1 python -m pytest $t -q -m gpu -p no:cacheprovider 2>&1 | tail -4
  wc -l $OUT
done
