mkdir -p gpurun_out
D=$PWD/shineon-virtual-tryon_amd
python -m pytest tests/test_ops_gpu.py -q -m gpu 2>&1 | tail -25 > gpurun_out/r05_c_ops_tests.txt
{
for lib in "" $D/libshineon_hip_old.so; do
  for shape in "8 256 192 64 64" "8 128 96 64 128" "8 128 96 128 128" "8 64 48 128 256" "4 256 192 64 64" "4 128 96 128 128"; do
    SHINEON_LIB=$lib python tools/one_layer.py wino $shape 3 1 1 40 2>&1 | grep -v amdgpu | sed "s|^|${lib:+old }|"
  done
done
python tools/ablate_bench.py new
} > gpurun_out/r05_c_ab.txt 2>&1
for c in c4 c3 c2; do
SHINEON_LIB=$D/libshineon_hip_old.so python bench.py --config $c --no-cpu-baseline --no-hbm-table > gpurun_out/r05_c_bench_${c}_old.json 2> /dev/null
python bench.py --config $c --no-cpu-baseline --no-hbm-table > gpurun_out/r05_c_bench_${c}_new.json 2> gpurun_out/r05_c_bench_${c}_new.log
done
cat gpurun_out/r05_c_ops_tests.txt
