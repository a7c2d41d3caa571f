cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SHINEON_AUTOTUNE=2 SHINEON_PLANS=$GRAFT_REPO_ROOT/shineon-virtual-tryon_amd/plans/gfx950.txt timeout 2400 python tools/make_plans.py gpurun_out/r03_j_plans.txt > gpurun_out/r03_j_make_plans.log 2>&1; echo "plans rc=$?"
tail -5 gpurun_out/r03_j_make_plans.log; wc -l gpurun_out/r03_j_plans.txt
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --plans gpurun_out/r03_j_plans.txt 2>/dev/null | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 --plans gpurun_out/r03_j_plans.txt 2>/dev/null | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c2 --plans gpurun_out/r03_j_plans.txt 2>/dev/null | cut -c1-200
