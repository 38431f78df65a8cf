#!/bin/bash
# build/var/prof/libpace_hip.so: the product objects with k_fvt.hip, k_riem3f.hip and k_dsw.hip replaced by tools/census/{fvt,riem,dsw}_prof.hip
# (stage stamps).  Extra compiler flags as arguments.
set -e
D=build/var/prof; mkdir -p $D
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
$CC "$@" -c tools/census/fvt_prof.hip -o $D/k_fvt.o
$CC "$@" -c tools/census/riem_prof.hip -o $D/k_riem3f.o
$CC "$@" -c tools/census/dsw_prof.hip -o $D/k_dsw.o
$CC "$@" -c tools/census/csw_prof.hip -o $D/k_csw.o
OBJS=""
for f in build/hip/*.o; do s=$(basename $f .o); if [ -f $D/$s.o ]; then OBJS="$OBJS $D/$s.o"; else OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $D/libpace_hip.so
echo built $D/libpace_hip.so
