#!/bin/bash
# Round-6 first GPU job: today's baseline numbers + the counters VERDICT r05 item 3 asks for (fused Winograd kernel: LDS
# instructions / bank conflicts / LDS issue stalls, L2 hit rate) + the list of ATen launches still inside a step.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python3 bench.py > $OUT/r06a_bench_c4.json 2> $OUT/r06a_bench_c4.log
python3 bench.py --config c2 --no-cpu-baseline > $OUT/r06a_bench_c2.json 2> $OUT/r06a_bench_c2.log
python3 tools/count_aten_ops.py > $OUT/r06a_aten_ops.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for LAYER in "8 256 192 64 64" "8 128 96 128 128" "4 128 96 128 64"; do
  TAGL=$(echo $LAYER | tr ' ' '_')
  for PASS in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA" \
              "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
              "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    NAME=$(echo $PASS | cut -d' ' -f1)
    rm -rf /tmp/prof_pmc
    timeout -k 5 240 rocprofv3 --kernel-trace --pmc $PASS -d /tmp/prof_pmc -o r -- python3 $R/tools/one_layer.py wino $LAYER 3 1 1 > $OUT/r06a_pmc_${TAGL}_$NAME.log 2>&1
    DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
    [ -n "$DB" ] && python3 $R/tools/rocpd_summary.py $DB $OUT/r06a_wino_${TAGL}_$NAME >> $OUT/r06a_pmc_summary.txt 2>&1
  done
done
cd $R
tail -1 $OUT/r06a_bench_c4.json | cut -c1-400; tail -1 $OUT/r06a_bench_c2.json | cut -c1-300
head -60 $OUT/r06a_aten_ops.txt
cat $OUT/r06a_pmc_summary.txt | head -80
