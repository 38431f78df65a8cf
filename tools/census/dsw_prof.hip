// Development tool (not product): k_dsw.hip compiled with wall-clock stamps (100 MHz, one clock for the whole device) in every
// workgroup of k_divdamp_fused: start, stage boundaries, end.  Built INSTEAD of k_dsw.hip into build/var/prof/libpace_hip.so by
// tools/build_prof.sh; read back with pace_debug_dd_prof (tools/dd_stage_times.py).
#include <hip/hip_runtime.h>
#define DD_PROF_BLOCKS 8192
__device__ long long g_dd_prof[DD_PROF_BLOCKS * 8];
#define DD_STAMP(n)                                                                                    \
  do {                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < DD_PROF_BLOCKS) g_dd_prof[blockIdx.x * 8 + (n)] = (long long)wall_clock64(); \
  } while (0)
#include "../../pace_amd/csrc/k_dsw.hip"
extern "C" int pace_debug_dd_prof(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dd_prof), sizeof(long long) * DD_PROF_BLOCKS * 8);
}
