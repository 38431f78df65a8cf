// Development tool: one kernel per (mode, tile variant) of the lean transport, so that the static instruction count of a
// kernel IS the dynamic count of that variant's straight-line path.  python tools/isa_census.py ../tools/census/fvt_variants
#include "../../pace_amd/csrc/fvt_core.h"
#if FVT_AVAILABLE
using namespace fvt;
template <int MORD, int DMODE, int EPI, bool EX, bool EY>
__global__ void __launch_bounds__(256, 4) k_var(Geo g, FvMet m, const real* __restrict__ q, const real* __restrict__ crx,
                                                const real* __restrict__ cry, const real* __restrict__ xfx,
                                                const real* __restrict__ yfx, real* __restrict__ fx, real* __restrict__ fy,
                                                const real* __restrict__ xunit, const real* __restrict__ yunit, FvDamp dp) {
  __shared__ FvtLds L;
  fvt_tile<MORD, DMODE, EPI, EX, EY>(L, g, m, q, crx, cry, xfx, yfx, fx, fy, xunit, yunit, dp, blockIdx.x, blockIdx.y, blockIdx.z);
}
#define INST(D, E, X, Y) template __global__ void k_var<6, D, E, X, Y>(Geo, FvMet, const real*, const real*, const real*, const real*, const real*, real*, real*, const real*, const real*, FvDamp);
#define INST4(D, E) INST(D, E, false, false) INST(D, E, true, false) INST(D, E, false, true) INST(D, E, true, true)
INST4(-1, 0)
INST4(1, 0)
INST4(0, 0)
INST4(3, 1)
INST4(2, 1)
#endif
#if FVT_AVAILABLE
template <int MORD, bool EX, bool EY>
__global__ void __launch_bounds__(256, 2) k_var_scalars(Geo g, FvMet m, FvtScalars S) {
  __shared__ FvtLdsScalars L;
  fvt_scalars_tile<MORD, EX, EY>(L, g, m, S, blockIdx.x, blockIdx.y, blockIdx.z);
}
template __global__ void k_var_scalars<6, false, false>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars<6, true, false>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars<6, false, true>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars<6, true, true>(Geo, FvMet, FvtScalars);
// the 512-thread form (x-runs and y-runs in different waves)
template <int MORD, bool EX, bool EY>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k_var_scalars_split(Geo g, FvMet m, FvtScalars S) {
  __shared__ FvtLdsScalars L;
  fvt_scalars_tile_split<MORD, EX, EY>(L, g, m, S, blockIdx.x, blockIdx.y, blockIdx.z);
}
template __global__ void k_var_scalars_split<6, false, false>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars_split<6, true, false>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars_split<6, false, true>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars_split<6, true, true>(Geo, FvMet, FvtScalars);
#endif
#if FVT_AVAILABLE
// the resident-layout form (round 6)
template <int MORD, bool EX, bool EY>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k_var_scalars_res(Geo g, FvMet m, FvtScalars S) {
  __shared__ FvtLdsScalarsRes L;
  fvt_scalars_tile_res<MORD, EX, EY>(L, g, m, S, blockIdx.x, blockIdx.y, blockIdx.z);
}
template __global__ void k_var_scalars_res<6, false, false>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars_res<6, true, false>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars_res<6, false, true>(Geo, FvMet, FvtScalars);
template __global__ void k_var_scalars_res<6, true, true>(Geo, FvMet, FvtScalars);
#endif
