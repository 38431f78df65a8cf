#!/bin/bash
# build/var/prof/libpace_hip.so: the product objects with k_fvt.hip replaced by tools/census/fvt_prof.hip (stage stamps).
set -e
D=build/var/prof; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $1 -c tools/census/fvt_prof.hip -o $D/k_fvt.o
OBJS=""
for f in build/hip/*.o; do s=$(basename $f .o); if [ "$s" = "k_fvt" ]; then OBJS="$OBJS $D/k_fvt.o"; else OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $D/libpace_hip.so
echo built $D/libpace_hip.so
