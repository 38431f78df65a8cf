// Development tool (not product): k_csw.hip compiled with wall-clock stamps (100 MHz, one clock for the whole device) in every
// workgroup of k_csw_tile: start, phase boundaries, end.  Built INSTEAD of k_csw.hip into build/var/prof/libpace_hip.so by
// tools/build_prof.sh; read back with pace_debug_csw_prof (tools/csw_stage_times.py).
#include <hip/hip_runtime.h>
#define CSW_PROF_BLOCKS 8192
__device__ long long g_csw_prof[CSW_PROF_BLOCKS * 8];
#define CSW_STAMP(n)                                                                                   \
  do {                                                                                                 \
    const unsigned b__ = blockIdx.x;                                          \
    if (threadIdx.x == 0 && b__ < CSW_PROF_BLOCKS) g_csw_prof[b__ * 8 + (n)] = (long long)wall_clock64(); \
  } while (0)
#include "../../pace_amd/csrc/k_csw.hip"
extern "C" int pace_debug_csw_prof(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_csw_prof), sizeof(long long) * CSW_PROF_BLOCKS * 8);
}
