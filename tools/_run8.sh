cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "1000 1" "64 2" "16 2" "4 2"; do set -- $cfg
  SHINEON_BUCKETED=1 SHINEON_BUCKET_MB=$1 SHINEON_BUCKETS_MIN=$2 timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 2> gpurun_out/r03_h_c3_b$1.log | cut -c1-200 | sed "s/^/bucket_mb=$1 min=$2: /"
done
timeout 300 python bench.py --no-cpu-baseline --no-hbm-table --config c3 2>/dev/null | cut -c1-200 | sed "s/^/plain: /"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt; timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-hbm-table > /tmp/kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/gpurun_out/r03_h_c4 gaps > $GRAFT_REPO_ROOT/gpurun_out/r03_h_c4_kernel_trace_summary.txt 2>&1
head -5 $GRAFT_REPO_ROOT/gpurun_out/r03_h_c4_kernel_trace_summary.txt
