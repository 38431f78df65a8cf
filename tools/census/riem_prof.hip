// Development tool (not product): k_riem3f.hip compiled with shader-clock stamps at the stage boundaries of the column solver --
// the first wave of workgroup 5 of every row.  Built INSTEAD of k_riem3f.hip into build/var/prof/libpace_hip.so by
// tools/build_prof.sh; read back with pace_debug_riem_prof (tools/riem_stage_times.py).
#include <hip/hip_runtime.h>
__device__ long long g_riem_prof[256 * 16];
#define RIEM_STAMP(n)                                                                                   \
  do {                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (threadIdx.x == 0 && blockIdx.x == 5 && blockIdx.y < 256)                                        \
      g_riem_prof[blockIdx.y * 16 + (n)] = (long long)__builtin_readcyclecounter();                     \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  } while (0)
#include "../../pace_amd/csrc/k_riem3f.hip"
extern "C" int pace_debug_riem_prof(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_riem_prof), sizeof(long long) * 256 * 16);
}
