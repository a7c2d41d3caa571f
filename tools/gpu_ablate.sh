mkdir -p gpurun_out
D=$PWD/shineon-virtual-tryon_amd
{
python tools/ablate_bench.py base
for m in 1 2 4 8 16 6 22 30; do SHINEON_LIB=$D/libshineon_hip_abl$m.so python tools/ablate_bench.py abl$m; done
} > gpurun_out/r05_igemm_ablation.txt 2>&1
sort -k2,5 -s gpurun_out/r05_igemm_ablation.txt | grep -v amdgpu.ids
