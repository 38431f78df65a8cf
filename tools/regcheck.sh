#!/bin/bash
# compile one source with flags and print per-kernel register / spill / LDS numbers:  tools/regcheck.sh "<flags>" [stem]
STEM=${2:-k_fvtp2d}
D=build/regcheck; mkdir -p $D; rm -f $D/*
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-inline-asm $1 -c pace_amd/csrc/$STEM.hip -o $D/x.o || exit 1
(cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $D/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | grep -E "\.name:|\.vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - | sed 's/ \+/ /g; s/_Z[0-9]*//; s/Ev3Geo.*FvDamp//'
