#!/bin/bash
# Builds measurement variants of the library with one ingredient of the igemm K loop removed (SO_ABLATE bit mask, see
# csrc/igemm2.hip) as shineon-virtual-tryon_amd/libshineon_hip_abl<mask>.so.  Results of those builds are WRONG by design.
set -e
cd "$(dirname "$0")/../shineon-virtual-tryon_amd/csrc"
make -j4 >/dev/null
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -DSO_ABLATE=$m -c igemm2.hip -o /tmp/igemm2_abl$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/igemm2_abl$m.o thin.o elementwise.o norm.o gmm.o dataprep.o sb16.o sams.o wino.o -o ../libshineon_hip_abl$m.so
done
ls -la ../libshineon_hip_abl*.so
