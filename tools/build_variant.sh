#!/bin/bash
# Experiment builds: recompile the named kernel sources with extra flags and link them with the product objects of the rest.
#   tools/build_variant.sh <name> "<extra flags>" [source stems ...]      (default source: k_fvtp2d)
# -> build/var/<name>/libpace_hip.so   (run `make` first: the other objects come from build/hip/)
set -e
NAME=$1; FLAGS=$2; shift 2
STEMS=${@:-k_fvtp2d}
D=build/var/$NAME
mkdir -p $D
OBJS=""
for f in build/hip/*.o; do
  s=$(basename $f .o)
  if [[ " $STEMS " == *" $s "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-inline-asm $FLAGS -c pace_amd/csrc/$s.hip -o $D/$s.o
    OBJS="$OBJS $D/$s.o"
  else
    OBJS="$OBJS $f"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $D/libpace_hip.so
echo built $D/libpace_hip.so
