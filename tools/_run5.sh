cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 tools/probes/stream_wait_value.bin > gpurun_out/r03_e_stream_wait_value.txt 2>&1; echo "probe rc=$?"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd" > gpurun_out/r03_e_tests_wino.log 2>&1; echo "wino tests rc=$?"
timeout 600 python tools/wino_bench.py > gpurun_out/r03_e_wino_bench.csv 2> gpurun_out/r03_e_wino_bench.err; echo "wino bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03_e_bench_c4.json 2> gpurun_out/r03_e_bench_c4.log; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
for L in "wgrad 4 16 12 512 512 3 1 1" "wino 8 256 192 64 64 3 1 1" "wino 8 64 48 256 256 3 1 1" "fprop 8 64 48 256 256 3 1 1"; do
  TAG=$(echo $L | tr ' ' '_')
  python3 $R/tools/one_layer.py $L > $R/gpurun_out/r03_e_one_$TAG.txt 2>&1
  for PASS in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
    NAME=$(echo $PASS | cut -d' ' -f1)
    rm -rf /tmp/prof_pmc
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $PASS -d /tmp/prof_pmc -o r -- python3 $R/tools/one_layer.py $L > /tmp/pmc.log 2>&1
    DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
    python3 $R/tools/rocpd_summary.py $DB $R/gpurun_out/r03_e_${TAG}_$NAME >> $R/gpurun_out/r03_e_pmc_summary.txt 2>&1
  done
done
cd $R
cat gpurun_out/r03_e_stream_wait_value.txt; tail -3 gpurun_out/r03_e_tests_wino.log; cat gpurun_out/r03_e_wino_bench.csv; cut -c1-200 gpurun_out/r03_e_bench_c4.json; cat gpurun_out/r03_e_one_*.txt; cat gpurun_out/r03_e_pmc_summary.txt | tail -20
