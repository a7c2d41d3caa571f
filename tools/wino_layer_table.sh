#!/bin/bash
# Per-layer table of the fused Winograd kernel on the VGG19 shapes of the try-on step (8 images forward, 4 images input
# gradient): us per launch, executed TFLOP/s (direct-convolution FLOPs / 2.25) and its fraction of the fp32 MFMA peak.
#   gpurun -- 'bash tools/wino_layer_table.sh > gpurun_out/wino_bench.csv'
cd ${GRAFT_REPO_ROOT:-.}
echo "batch,H,W,C,Ko,us_per_launch,direct_equivalent_tflops,executed_tflops,executed_frac_of_157.3"
for S in "8 256 192 64 64" "8 128 96 64 128" "8 128 96 128 128" "8 64 48 128 256" "4 256 192 64 64" "4 128 96 128 64" "4 128 96 128 128" "4 64 48 256 128"; do
  python3 tools/one_layer.py wino $S 3 1 1 300 2>/dev/null | tail -1 | python3 -c "
import sys,re
l=sys.stdin.read()
us=float(re.search(r'([0-9.]+) us/launch',l).group(1)); tf=float(re.search(r'([0-9.]+) TFLOP/s',l).group(1))
print('$S'.replace(' ',',')+',%.1f,%.1f,%.1f,%.3f'%(us,tf,tf/2.25,tf/2.25/157.3))"
done
