#!/bin/bash
# Regenerates the judged profile artefacts on a GPU box:  tools/profile_round.sh TAG   (run from the repo root via gpurun)
# Outputs land in gpurun_out/ (merged back by gpurun); copy the ones to keep into profiles/ and run tools/make_traffic.py per
# configuration (see profiles/README.md).  Under rocprofv3 the python program goes directly after `--`.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
# 1. bench lines (graph replay, CPU baseline on) + per-launch timing dump of the instrumented eager pass
for CFG in c4 c1 c2 c3 c5 sams; do
  SO_PROF_DUMP=$OUT/${TAG}_${CFG}_igemm_launches.csv python3 bench.py --config $CFG > $OUT/${TAG}_bench_${CFG}.json 2> $OUT/${TAG}_bench_${CFG}.log
done
# 1b. sustained runs: >= 2 s of timed steps per configuration (the default 20-step window is 0.13 s on c4)
for CS in "c4 400" "c1 2500" "c2 1200" "c3 500" "c5 200" "sams 5"; do
  set -- $CS
  python3 bench.py --config $1 --steps $2 --no-cpu-baseline --no-hbm-table 2> /dev/null | grep '^{' > $OUT/${TAG}_sustained_$1.json
done
# 2. kernel trace of the headline command
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_c4 gaps > $OUT/${TAG}_kernel_trace_summary.txt 2>&1
python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_c4 seq 1100 >> $OUT/${TAG}_kernel_trace_summary.txt 2>&1
# 3. PMC passes (own runs, kernel-trace only): HBM read, HBM write per configuration; MFMA / SQ activity for the headline.
#    Eager launches (--no-graph).  SAMS: the whole step and a bs = 1 slice both die under the counters (rounds 2 and 3), so its
#    dominant kernel - the fused Winograd kernel on its most frequent layer, 128 -> 256 channels at 256x192, bs = 4 - is
#    counted in a single-layer process (tools/one_layer.py).
for PASS in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $PASS -d /tmp/prof_pmc -o r -- python3 $R/tools/one_layer.py wino 4 256 192 128 256 3 1 1 > $OUT/${TAG}_pmc_sams_layer_$PASS.log 2>&1
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_sams_wino_fused_$PASS >> $OUT/${TAG}_kernel_trace_summary.txt 2>&1
done
# SAMS, the REAL step (the timed batch bs = 4, one warm-up + one step) with the counters restricted to its dominant kernel (--kernel-include-regex): the
# unfiltered passes of rounds 2 and 3 killed the process; if this one survives, traffic.json [sams] comes from the step itself
for PASS in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $PASS --kernel-include-regex "wino_fused_k" -d /tmp/prof_pmc -o r -- python3 $R/bench.py --config sams --batch 4 --steps 1 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-hbm-table > $OUT/${TAG}_pmc_sams_step_$PASS.log 2>&1
  echo "sams step pmc $PASS rc=$?" >> $OUT/${TAG}_kernel_trace_summary.txt
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  [ -n "$DB" ] && python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_sams_step_$PASS >> $OUT/${TAG}_kernel_trace_summary.txt 2>&1
done
for CFG in c4 c2 c3; do
  EXTRA="--steps 3 --warmup 1 --no-graph --ramp-seconds 0"
  PASSES=("FETCH_SIZE" "WRITE_SIZE")
  [ $CFG = c4 ] && PASSES+=("SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE")
  for PASS in "${PASSES[@]}"; do
    NAME=$(echo $PASS | cut -d' ' -f1)
    rm -rf /tmp/prof_pmc
    timeout -k 5 900 rocprofv3 --kernel-trace --pmc $PASS -d /tmp/prof_pmc -o r -- python3 $R/bench.py --config $CFG $EXTRA --no-cpu-baseline --no-hbm-table > $OUT/${TAG}_pmc_${CFG}_$NAME.log 2>&1
    DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
    python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_${CFG}_$NAME >> $OUT/${TAG}_kernel_trace_summary.txt 2>&1
  done
done
cd $R
# 4. the exchange path over RCCL itself: a ONE-rank nccl group (every collective issued, nothing on the wire)
for CFG in c4 c3 sams; do
  SHINEON_SINGLE_RANK_GROUP=1 timeout -k 5 600 python3 bench.py --config $CFG --no-cpu-baseline --no-hbm-table 2> $OUT/${TAG}_single_rank_rccl_${CFG}.log | grep '^{' > $OUT/${TAG}_bench_${CFG}_single_rank_rccl.json
  grep "exposed\|timed\|backend" $OUT/${TAG}_single_rank_rccl_${CFG}.log
done
for CFG in c4 c1 c2 c3 c5 sams; do tail -1 $OUT/${TAG}_bench_${CFG}.json | cut -c1-300; done
for CFG in c4 c1 c2 c3 c5 sams; do echo "sustained $CFG: $(cut -c1-200 $OUT/${TAG}_sustained_${CFG}.json)"; done
cat $OUT/${TAG}_kernel_trace_summary.txt
