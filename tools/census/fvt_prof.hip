// Development tool (not product): k_fvt.hip compiled with shader-clock stamps at the stage boundaries of the scalar-phase
// kernel -- one interior, one corner, one west-edge and one south-edge workgroup per level.  Built INSTEAD of k_fvt.hip into
// build/var/prof/libpace_hip.so by tools/build_prof.sh; read back with pace_debug_fvt_prof (tools/fvt_stage_times.py).
#include <hip/hip_runtime.h>
__device__ long long g_fvt_prof[4 * 128 * 32];
#define FVT_SLOT(bx, by) (((bx) == 2 && (by) == 3) ? 0 : ((bx) == 0 && (by) == 0) ? 1 : ((bx) == 0 && (by) == 3) ? 2 : ((bx) == 2 && (by) == 0) ? 3 : -1)
#define FVT_STAMP(n)                                                                     \
  do {                                                                                   \
    if (threadIdx.x == 0 && k < 128 && FVT_SLOT(bx, by) >= 0)                            \
      g_fvt_prof[(FVT_SLOT(bx, by) * 128 + k) * 32 + (n)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
__device__ long long g_fvt_arrive[4 * 128 * 64];
#define FVT_ARRIVE(n)                                                                    \
  do {                                                                                   \
    if ((threadIdx.x == 0 || threadIdx.x == 256) && k < 128 && FVT_SLOT(bx, by) >= 0)    \
      g_fvt_arrive[(FVT_SLOT(bx, by) * 128 + k) * 64 + 2 * (n) + (threadIdx.x == 256)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#include "../../pace_amd/csrc/k_fvt.hip"
extern "C" int pace_debug_fvt_arrive(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fvt_arrive), sizeof(long long) * 4 * 128 * 64);
}
extern "C" int pace_debug_fvt_prof(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fvt_prof), sizeof(long long) * 4 * 128 * 32);
}
