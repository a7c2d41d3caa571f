#!/bin/bash
# Measurement variants of the fused Winograd kernel with one ingredient of its K step removed (WF_ABLATE bit mask, see
# csrc/wino.hip) as shineon-virtual-tryon_amd/libshineon_hip_wabl<mask>.so.  Results of those builds are WRONG by design.
set -e
cd "$(dirname "$0")/../shineon-virtual-tryon_amd/csrc"
make -j4 >/dev/null
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -Xclang -target-feature -Xclang -packed-fp32-ops -DWF_ABLATE=$m -c wino.hip -o /tmp/wino_abl$m.o &
done
wait
# every other object of the library, from the Makefile's own source list (a hand-written list once missed pgemm.o)
OTHERS=$(sed -n 's/^SRCS = //p' Makefile | tr ' ' '\n' | sed 's/\.hip$/.o/' | grep -v '^wino\.o$' | tr '\n' ' ')
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/wino_abl$m.o -o ../libshineon_hip_wabl$m.so
done
ls ../libshineon_hip_wabl*.so
