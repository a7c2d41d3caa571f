mkdir -p gpurun_out
D=$PWD/shineon-virtual-tryon_amd
{
for m in "" 1 2 4 8 16 32 6 30; do
  for shape in "8 256 192 64 64" "8 128 96 128 128" "8 64 48 128 256"; do
    SHINEON_LIB=${m:+$D/libshineon_hip_wabl$m.so} python tools/one_layer.py wino $shape 3 1 1 40 2>&1 | grep -v amdgpu | sed "s|^|abl${m:-0} |"
  done
done
} > gpurun_out/r05_wino_ablation.txt 2>&1
cat gpurun_out/r05_wino_ablation.txt
