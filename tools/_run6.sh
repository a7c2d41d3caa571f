cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( for L in "wgrad 4 16 12 512 512 3 1 1" "fprop 4 16 12 512 512 3 1 1" "dgrad 4 16 12 512 512 3 1 1" "wgrad 4 64 48 128 256 4 2 1" "fprop 4 64 48 128 256 4 2 1" "dgrad 4 64 48 128 256 4 2 1" "wgrad 4 32 24 256 512 4 2 1" "fprop 4 32 24 256 512 4 2 1" "wino 8 256 192 64 64 3 1 1" "wino 8 128 96 128 128 3 1 1"; do python3 tools/one_layer.py $L; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_f_one_layers.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03_f_bench_c4.json 2> gpurun_out/r03_f_bench_c4.log; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --config c2 > gpurun_out/r03_f_bench_c2.json 2> gpurun_out/r03_f_bench_c2.log; echo "bench c2 rc=$?"
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q > gpurun_out/r03_f_tests.log 2>&1; echo "tests rc=$?"
cat gpurun_out/r03_f_one_layers.txt; cut -c1-200 gpurun_out/r03_f_bench_c4.json; cut -c1-200 gpurun_out/r03_f_bench_c2.json; tail -3 gpurun_out/r03_f_tests.log
