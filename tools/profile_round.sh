#!/bin/bash
# Regenerates the judged profile artefacts on a GPU box:  tools/profile_round.sh TAG   (run from the repo root via gpurun)
# Outputs land in gpurun_out/ (merged back by gpurun); copy the ones to keep into profiles/.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
PLANS=$OUT/plans_$TAG.txt
cd $R
# 1. default bench line (graph replay, CPU baseline on) + per-launch timing dump of the instrumented eager pass
SO_PROF_DUMP=$OUT/${TAG}_igemm_launches.csv python3 bench.py --plans $PLANS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.log
# 2. kernel trace of the same command (plans preloaded so no tuning launches pollute the statistics)
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python3 $R/bench.py --no-cpu-baseline --plans $PLANS > $OUT/${TAG}_kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG} gaps > $OUT/${TAG}_kernel_trace_summary.txt 2>&1
# 3. PMC passes (own runs, kernel-trace only): HBM read, HBM write, MFMA/SQ activity.  Eager launches (--no-graph).
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  NAME=$(echo $PASS | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $PASS -d /tmp/prof_pmc -o r -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --plans $PLANS > $OUT/${TAG}_pmc_$NAME.log 2>&1
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_$NAME >> $OUT/${TAG}_kernel_trace_summary.txt 2>&1
done
tail -1 $OUT/${TAG}_bench.json | cut -c1-400
cat $OUT/${TAG}_kernel_trace_summary.txt
