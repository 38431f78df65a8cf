// CPU emulation of the small slice of the HIP programming model our kernels use.
//
// TEST INFRASTRUCTURE ONLY.  It lets the *same kernel sources* (pace_amd/csrc/*.hip) be compiled
// by g++ (optionally with -fsanitize=address,undefined -- GPU sanitizers are unavailable on the
// MI355X pool) and run inside the CPU-only container, so index logic, LDS tiling and barrier
// structure can be checked against the oracle before a kernel ever reaches a GPU.  The product
// loader (pace_amd/_lib.py) never loads the emulation library; see tests/emu/README.md.
//
// Model: blocks run one after another; the threads of a block are ucontext fibers that switch at
// __syncthreads(); __shared__ variables become function-local statics (shared by all fibers).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <ucontext.h>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
static inline int __mul24(int a, int b) { return a * b; }
#define __forceinline__ inline
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)
#define PACE_EMU 1

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }

void __syncthreads();
inline void __threadfence_block() {}
inline void __threadfence() {}
// Per workgroup: its threads run as plain calls until one reaches __syncthreads(), then that workgroup is run again with
// fibers; that is safe because nothing but LDS is written before a workgroup's first barrier.  (`mode`: unused, kept for the
// launch macro.)
void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body, int* mode, const char* name);

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                      \
  do {                                                                                    \
    static int emu_mode__ = 0;                                                            \
    emu_launch((grid), (block), [&]() { kernel(__VA_ARGS__); }, &emu_mode__, #kernel);    \
  } while (0)

// Wave shuffles (wave = 64 consecutive threads of the block, `width`-lane groups as in HIP): every lane deposits its value,
// all fibers are switched once, every lane reads its source lane's deposit, all fibers are switched again (so that the
// next shuffle cannot overwrite a deposit that has not been read yet).  Needs fibers, like __syncthreads().
void emu_shuffle_exchange(const void* mine, void* out, int src_lane_in_wave, int nbytes);
template <class T>
static inline T emu_shfl_from(T v, int src) {
  T out;
  emu_shuffle_exchange(&v, &out, src, (int)sizeof(T));
  return out;
}
static inline int emu_lane() { return (int)((threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63u); }
template <class T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
  const int lane = emu_lane(), idx = lane % width;
  return emu_shfl_from(v, idx >= (int)d ? lane - (int)d : lane);
}
template <class T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
  const int lane = emu_lane(), idx = lane % width;
  return emu_shfl_from(v, idx + (int)d < width ? lane + (int)d : lane);
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
  const int lane = emu_lane();
  return emu_shfl_from(v, (lane / width) * width + (src % width));
}

using std::exp; using std::log; using std::sqrt; using std::fabs; using std::fmin; using std::fmax;
using std::sin; using std::cos; using std::asin; using std::pow; using std::frexp; using std::ldexp; using std::rint; using std::fma;
