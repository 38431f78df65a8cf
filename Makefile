# Build recipes.  `make` builds the gfx950 shared library (hipcc cross-compiles without a GPU);
# `make emu` builds the CPU emulation of the same kernel sources (TEST INFRASTRUCTURE ONLY, see
# tests/emu/README.md).
HIPCC ?= /opt/rocm/bin/hipcc
CSRC := pace_amd/csrc
SRCS := $(CSRC)/capi.hip $(CSRC)/k_fxadv.hip $(CSRC)/k_fvtp2d.hip $(CSRC)/k_fvt.hip $(CSRC)/k_fvt16.hip $(CSRC)/k_delnflux.hip $(CSRC)/k_dsw.hip $(CSRC)/k_riem3.hip $(CSRC)/k_riem3f.hip $(CSRC)/k_sim1.hip $(CSRC)/k_ppm.hip $(CSRC)/k_csw.hip $(CSRC)/k_acoustic.hip $(CSRC)/k_halo.hip $(CSRC)/k_tracer.hip $(CSRC)/k_remap.hip $(CSRC)/k_l2e.hip $(CSRC)/k_dycore.hip $(CSRC)/k_stencils.hip
HDRS := $(CSRC)/k_fvt.hip $(CSRC)/common.h $(CSRC)/kernels.h $(CSRC)/delnflux_core.h $(CSRC)/fvt_core.h include/pace_hip.h
# -ffp-contract=off: no FMA contraction, so horizontal stencils are bit-comparable with the numpy oracle.
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function
OBJS := $(patsubst $(CSRC)/%.hip,build/hip/%.o,$(SRCS))
# the C ABI (pace_*) is all a library exports: $(CSRC)/exports.map
EXPORTS := -Wl,--version-script=$(CSRC)/exports.map

all: pace_amd/libpace_hip.so pace_amd/libpace_hip_f32.so

# the float32-storage build (BASELINE configuration 5): the same sources with pace_real_t = float
F32OBJS := $(patsubst $(CSRC)/%.hip,build/hip_f32/%.o,$(SRCS))
build/hip_f32/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/hip_f32
	$(HIPCC) $(HIPFLAGS) -DPACE_REAL_FLOAT -c $< -o $@
f32: pace_amd/libpace_hip_f32.so
pace_amd/libpace_hip_f32.so: $(F32OBJS) $(CSRC)/exports.map
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(EXPORTS) $(F32OBJS) -o $@

build/hip/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/hip
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

pace_amd/libpace_hip.so: $(OBJS) $(CSRC)/exports.map
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(EXPORTS) $(OBJS) -o $@

EMU_TI ?= 64
EMU_TJ ?= 16
# -fno-gnu-unique: the emulated __shared__ statics of template kernels must NOT be unified process-wide (STB_GNU_UNIQUE symbols
# are, even under RTLD_LOCAL): two emulation libraries with different tile constants would share LDS arrays of different sizes
EMUFLAGS := -O2 -g -std=c++17 -fPIC -fno-gnu-unique -ffp-contract=off -DPACE_EMU -Itests/emu -x c++ $(EMU_EXTRA)
EMUOBJS := $(patsubst $(CSRC)/%.hip,build/emu/%.o,$(SRCS))

build/emu/%.o: $(CSRC)/%.hip $(HDRS) tests/emu/hip_emu.h
	@mkdir -p build/emu
	g++ $(EMUFLAGS) -c $< -o $@

build/emu/hip_emu.o: tests/emu/hip_emu.cpp tests/emu/hip_emu.h
	@mkdir -p build/emu
	g++ -O2 -g -std=c++17 -fPIC -fno-gnu-unique -Itests/emu -c $< -o $@

emu: tests/emu/libpace_emu.so
tests/emu/libpace_emu.so: $(EMUOBJS) build/emu/hip_emu.o
	g++ -shared -fPIC $(EXPORTS) $(EMUOBJS) build/emu/hip_emu.o -o $@

EMUF32OBJS := $(patsubst $(CSRC)/%.hip,build/emu_f32/%.o,$(SRCS))
build/emu_f32/%.o: $(CSRC)/%.hip $(HDRS) tests/emu/hip_emu.h
	@mkdir -p build/emu_f32
	g++ $(EMUFLAGS) -DPACE_REAL_FLOAT -c $< -o $@
emu-f32: tests/emu/libpace_emu_f32.so
tests/emu/libpace_emu_f32.so: $(EMUF32OBJS) build/emu/hip_emu.o
	g++ -shared -fPIC $(EXPORTS) $(EMUF32OBJS) build/emu/hip_emu.o -o $@

# the same, with 4 x 4 transport / damping tiles and runs of 2 interfaces: at C12 this gives workgroups that touch no tile
# edge, so the straight-line interior code paths and every tile seam are exercised by the CPU test-suite as well
SMALLFLAGS := $(EMUFLAGS) -DFV_TI=4 -DFV_TJ=4 -DDN_TI=4 -DDN_TJ=4 -DFV_RF=2 -DDD_TI=5 -DDD_TJ=4 -DAB_TI=6 -DAB_TJ=3 -DCSW_TI=4 -DCSW_TJ=3
SMALLOBJS := $(patsubst $(CSRC)/%.hip,build/emu_small/%.o,$(SRCS))
build/emu_small/%.o: $(CSRC)/%.hip $(HDRS) tests/emu/hip_emu.h
	@mkdir -p build/emu_small
	g++ $(SMALLFLAGS) -c $< -o $@
emu-small: tests/emu/libpace_emu_small.so
tests/emu/libpace_emu_small.so: $(SMALLOBJS) build/emu/hip_emu.o
	g++ -shared -fPIC $(EXPORTS) $(SMALLOBJS) build/emu/hip_emu.o -o $@

# the same, with 8 x 8 transport tiles and runs of 3 interfaces: at C16 / C24 every tile edge coincides with a tile boundary of
# the workgroup tiling AND the run geometry satisfies ppm_run_canon's conditions (AHI = 2), so the CPU test-suite exercises the
# canonical edge path the C96 ... C384 production tiling takes
CANONFLAGS := $(EMUFLAGS) -DFV_TI=8 -DFV_TJ=8 -DDN_TI=8 -DDN_TJ=8 -DFV_RF=3
CANONOBJS := $(patsubst $(CSRC)/%.hip,build/emu_canon/%.o,$(SRCS))
build/emu_canon/%.o: $(CSRC)/%.hip $(HDRS) tests/emu/hip_emu.h
	@mkdir -p build/emu_canon
	g++ $(CANONFLAGS) -c $< -o $@
emu-canon: tests/emu/libpace_emu_canon.so
tests/emu/libpace_emu_canon.so: $(CANONOBJS) build/emu/hip_emu.o
	g++ -shared -fPIC $(EXPORTS) $(CANONOBJS) build/emu/hip_emu.o -o $@

clean:
	rm -rf build pace_amd/libpace_hip.so pace_amd/libpace_hip_f32.so tests/emu/libpace_emu.so tests/emu/libpace_emu_small.so tests/emu/libpace_emu_f32.so tests/emu/libpace_emu_canon.so

.PHONY: all f32 emu emu-f32 emu-small emu-canon clean
