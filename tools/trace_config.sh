#!/bin/bash
# Kernel trace of one bench configuration: per-kernel stats, exclusive attribution and one step as an ordered launch list.
#   tools/trace_config.sh TAG CONFIG N_LAST   (on a GPU box via gpurun; outputs in gpurun_out/<TAG>_<CONFIG>_*)
TAG=${1:-rXX}; CFG=${2:-c1}; NL=${3:-300}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tc
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tc -o r -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-hbm-table --ramp-seconds 0 --steps 40 > $OUT/${TAG}_${CFG}_trace.log 2>&1
DB=$(find /tmp/prof_tc -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_${CFG} gaps
python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_${CFG} seq $NL
tail -1 $OUT/${TAG}_${CFG}_trace.log | cut -c1-200
