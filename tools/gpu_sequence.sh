#!/bin/bash
# One steady-state step of the headline configuration as an ordered launch list:  tools/gpu_sequence.sh TAG [N_LAST]
TAG=${1:-rXX}; NL=${2:-540}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_seq
timeout -k 5 600 rocprofv3 --kernel-trace -d /tmp/prof_seq -o r -- python3 $R/bench.py --no-cpu-baseline --no-hbm-table --steps 6 --warmup 2 > $OUT/${TAG}_seq.log 2>&1
DB=$(find /tmp/prof_seq -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB $OUT/${TAG}_c4 seq $NL
tail -2 $OUT/${TAG}_seq.log | cut -c1-400
