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
# every other object of the library, from the Makefile's own source list (a hand-written list once missed pgemm.o)
OTHERS=$(sed -n 's/^SRCS = //p' Makefile | tr ' ' '\n' | sed 's/\.hip$/.o/' | grep -v '^igemm2\.o$' | tr '\n' ' ')
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/igemm2_abl$m.o $OTHERS -o ../libshineon_hip_abl$m.so
done
ls -la ../libshineon_hip_abl*.so
