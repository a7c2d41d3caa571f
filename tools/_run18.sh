cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SHINEON_AUTOTUNE=2 SHINEON_PLANS=$GRAFT_REPO_ROOT/shineon-virtual-tryon_amd/plans/gfx950.txt timeout 2400 python tools/make_plans.py gpurun_out/r03_r_plans.txt > gpurun_out/r03_r_make_plans.log 2>&1; echo "plans rc=$?"
tail -3 gpurun_out/r03_r_make_plans.log; wc -l gpurun_out/r03_r_plans.txt
cp gpurun_out/r03_r_plans.txt shineon-virtual-tryon_amd/plans/gfx950.txt
bash tools/profile_round.sh r03_s > gpurun_out/r03_s_profile_round.log 2>&1
tail -12 gpurun_out/r03_s_profile_round.log | cut -c1-300
timeout 600 python bench.py --config sams > gpurun_out/r03_s_bench_sams.json 2> gpurun_out/r03_s_bench_sams.log; cut -c1-300 gpurun_out/r03_s_bench_sams.json
timeout 600 python bench.py --config c5 > gpurun_out/r03_s_bench_c5.json 2> gpurun_out/r03_s_bench_c5.log; cut -c1-300 gpurun_out/r03_s_bench_c5.json
