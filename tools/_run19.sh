cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r03_t > gpurun_out/r03_t_profile_round.log 2>&1
tail -6 gpurun_out/r03_t_profile_round.log | cut -c1-200
SO_PROF_DUMP=gpurun_out/r03_t_sams_igemm_launches.csv timeout 900 python bench.py --config sams > gpurun_out/r03_t_bench_sams.json 2> gpurun_out/r03_t_bench_sams.log; cut -c1-200 gpurun_out/r03_t_bench_sams.json
timeout 600 python bench.py --config c5 > gpurun_out/r03_t_bench_c5.json 2> gpurun_out/r03_t_bench_c5.log; cut -c1-200 gpurun_out/r03_t_bench_c5.json
timeout 600 python bench.py --batch 8 --no-cpu-baseline --no-hbm-table > gpurun_out/r03_t_bench_c4_bs8.json 2> gpurun_out/r03_t_bench_c4_bs8.log; cut -c1-200 gpurun_out/r03_t_bench_c4_bs8.json
cd /tmp && export TMPDIR=/tmp
for L in "wgrad 4 128 96 128 256 3 1 1"; do
  TAG=sams_wgrad_128x128w8
  for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/prof_pmc
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $PASS -d /tmp/prof_pmc -o r -- python3 $GRAFT_REPO_ROOT/tools/one_layer.py wgrad 20 256 192 128 256 3 1 1 5 > /tmp/pmc.log 2>&1
    DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
    python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/gpurun_out/r03_t_${TAG}_$PASS > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu --durations=8 > gpurun_out/r03_t_tests_full.log 2>&1; echo "full gpu tests rc=$?"
tail -14 gpurun_out/r03_t_tests_full.log
