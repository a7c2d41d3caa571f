#!/bin/bash
# The bench lines of every configuration (defaults, CPU baseline on) -> gpurun_out/<TAG>_bench_<cfg>.json:  tools/bench_lines.sh TAG
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for CFG in c4 c1 c2 c3 c5 sams; do
  SO_PROF_DUMP=$OUT/${TAG}_${CFG}_igemm_launches.csv python3 bench.py --config $CFG > $OUT/${TAG}_bench_${CFG}.json 2> $OUT/${TAG}_bench_${CFG}.log
  tail -1 $OUT/${TAG}_bench_${CFG}.json | cut -c1-200
done
