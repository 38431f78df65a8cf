// Machine characterisation for the fp64 stencil kernels (development tool, not product): issue cost and latency of the
// instruction classes the transport kernel is made of, at 1 / 2 / 4 waves per SIMD, measured with the shader clock inside the
// kernel.  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/ubench.hip -o build/ubench ; run on the GPU box.
// Output: one line per (test, waves per SIMD): cycles per wave-instruction seen by ONE wave, and per SIMD (= that / waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e__ = (x);                                                               \
    if (e__ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__));       \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

#define REP8(X) X X X X X X X X
#define REP4(X) X X X X

struct Stamp {
  long long cyc, real;
};

// one block per CU is forced by the dynamic LDS size (> half of 160 KB)
#define PROLOG                                                      \
  extern __shared__ double lds[];                                   \
  const int tid = threadIdx.x;                                      \
  double a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7; \
  double b = seed[0];                                               \
  int i0 = tid, i1 = tid + 1, i2 = tid + 2, i3 = tid + 3, i4 = tid + 4, i5 = tid + 5, i6 = tid + 6, i7 = tid + 7; \
  int ib = (int)seed[1];                                            \
  (void)lds; (void)a0; (void)a1; (void)a2; (void)a3; (void)a4; (void)a5; (void)a6; (void)a7; (void)b; \
  (void)i0; (void)i1; (void)i2; (void)i3; (void)i4; (void)i5; (void)i6; (void)i7; (void)ib; \
  __syncthreads();                                                  \
  const long long r0 = wall_clock64();                              \
  const long long t0 = clock64();

#define EPILOG(NINSTR)                                              \
  const long long t1 = clock64();                                   \
  const long long r1 = wall_clock64();                              \
  if ((tid & 63) == 0) {                                            \
    const int w = blockIdx.x * (blockDim.x / 64) + tid / 64;        \
    st[w].cyc = t1 - t0;                                            \
    st[w].real = r1 - r0;                                           \
  }                                                                 \
  out[blockIdx.x * blockDim.x + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);

#define KERNEL(name) __global__ void __launch_bounds__(1024) name(const double* seed, double* out, Stamp* st, int iters)

// ---- fp64 -------------------------------------------------------------------------------------------------------
KERNEL(k_f64_add_dep) {  // one dependent chain: latency of v_add_f64
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile(REP8("v_add_f64 %0, %0, %1\n") : "+v"(a0) : "v"(b));
  }
  EPILOG(8)
}
KERNEL(k_f64_add_ind) {  // eight independent chains: issue cost
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\n"
                 "v_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b));
  }
  EPILOG(8)
}
KERNEL(k_f64_mul_ind) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_mul_f64 %0, %0, %8\nv_mul_f64 %1, %1, %8\nv_mul_f64 %2, %2, %8\nv_mul_f64 %3, %3, %8\n"
                 "v_mul_f64 %4, %4, %8\nv_mul_f64 %5, %5, %8\nv_mul_f64 %6, %6, %8\nv_mul_f64 %7, %7, %8\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b));
  }
  EPILOG(8)
}
KERNEL(k_f64_fma_ind) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\n"
                 "v_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b));
  }
  EPILOG(8)
}
KERNEL(k_f64_rcp_ind) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3\n"
                 "v_rcp_f64 %4, %4\nv_rcp_f64 %5, %5\nv_rcp_f64 %6, %6\nv_rcp_f64 %7, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  EPILOG(8)
}
KERNEL(k_f64_div) {  // the compiler's IEEE division sequence, eight independent quotients per iteration
  PROLOG
  for (int it = 0; it < iters; ++it) {
    a0 = a0 / b; a1 = a1 / b; a2 = a2 / b; a3 = a3 / b; a4 = a4 / b; a5 = a5 / b; a6 = a6 / b; a7 = a7 / b;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  EPILOG(8)
}
// ---- 32-bit VALU ------------------------------------------------------------------------------------------------
KERNEL(k_i32_add_dep) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile(REP8("v_add_u32 %0, %0, %1\n") : "+v"(i0) : "v"(ib));
  }
  EPILOG(8)
}
KERNEL(k_i32_add_ind) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                 "v_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                 : "v"(ib));
  }
  EPILOG(8)
}
KERNEL(k_i32_mad24_ind) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_mad_i32_i24 %0, %0, %8, %8\nv_mad_i32_i24 %1, %1, %8, %8\nv_mad_i32_i24 %2, %2, %8, %8\nv_mad_i32_i24 %3, %3, %8, %8\n"
                 "v_mad_i32_i24 %4, %4, %8, %8\nv_mad_i32_i24 %5, %5, %8, %8\nv_mad_i32_i24 %6, %6, %8, %8\nv_mad_i32_i24 %7, %7, %8, %8\n"
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                 : "v"(ib));
  }
  EPILOG(8)
}
KERNEL(k_cndmask_ind) {  // a 64-bit select = two of these
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_cmp_lt_i32 vcc, %0, %8\n"
                 "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                 "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\n"
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                 : "v"(ib)
                 : "vcc");
  }
  EPILOG(8)
}
KERNEL(k_mix_f64_i32) {  // 4 fp64 adds interleaved with 4 integer adds
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_add_f64 %0, %0, %8\nv_add_u32 %4, %4, %9\nv_add_f64 %1, %1, %8\nv_add_u32 %5, %5, %9\n"
                 "v_add_f64 %2, %2, %8\nv_add_u32 %6, %6, %9\nv_add_f64 %3, %3, %8\nv_add_u32 %7, %7, %9\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3)
                 : "v"(b), "v"(ib));
  }
  EPILOG(8)
}
KERNEL(k_mix_valu_salu) {  // 4 fp64 adds interleaved with 4 scalar adds
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_add_f64 %0, %0, %4\ns_add_u32 s40, s40, s41\nv_add_f64 %1, %1, %4\ns_add_u32 s41, s41, s40\n"
                 "v_add_f64 %2, %2, %4\ns_add_u32 s40, s40, s41\nv_add_f64 %3, %3, %4\ns_add_u32 s41, s41, s40\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                 : "v"(b)
                 : "s40", "s41", "scc");
  }
  EPILOG(8)
}
KERNEL(k_salu_ind) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_add_u32 s40, s40, s41\ns_add_u32 s41, s41, s42\ns_add_u32 s42, s42, s43\ns_add_u32 s43, s43, s40\n"
                 "s_add_u32 s40, s40, s41\ns_add_u32 s41, s41, s42\ns_add_u32 s42, s42, s43\ns_add_u32 s43, s43, s40\n"
                 ::: "s40", "s41", "s42", "s43", "scc");
  }
  EPILOG(8)
}
KERNEL(k_dpp_shr) {  // fp64 value shifted across the wave by one lane: two v_mov_b32 with wave_shr:1
  PROLOG
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 "v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 "v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 "v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 "v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 "v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 "v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
  }
  EPILOG(8)
}
// ---- LDS --------------------------------------------------------------------------------------------------------
KERNEL(k_lds_read_dep) {  // pointer chase through LDS: latency of ds_read_b32
  PROLOG
  int* li = (int*)lds;
  for (int e = tid; e < 4096; e += blockDim.x) li[e] = (e + 64) & 4095;
  __syncthreads();
  int p = tid & 4095;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p = li[p];
  }
  i0 += p;
  EPILOG(8)
}
KERNEL(k_lds_read64_ind) {  // eight independent conflict-free ds_read_b64 per wait
  PROLOG
  for (int e = tid; e < 8192; e += blockDim.x) lds[e] = e;
  __syncthreads();
  const double* p = lds + tid;
  for (int it = 0; it < iters; ++it) {
    double x0 = p[0], x1 = p[1024], x2 = p[2048], x3 = p[3072], x4 = p[4096], x5 = p[5120], x6 = p[6144], x7 = p[7168];
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    a0 += x0; a1 += x1; a2 += x2; a3 += x3; a4 += x4; a5 += x5; a6 += x6; a7 += x7;
    asm volatile("" ::: "memory");
  }
  EPILOG(16)
}
KERNEL(k_lds_write64_ind) {
  PROLOG
  double* p = lds + tid;
  for (int it = 0; it < iters; ++it) {
    p[0] = a0; p[1024] = a1; p[2048] = a2; p[3072] = a3; p[4096] = a4; p[5120] = a5; p[6144] = a6; p[7168] = a7;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  a0 += lds[(tid * 7) & 8191];
  EPILOG(8)
}
KERNEL(k_barrier) {
  PROLOG
  for (int it = 0; it < iters; ++it) {
    REP8(__syncthreads();)
  }
  EPILOG(8)
}
// ---- global memory ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_gl_chase(const int* __restrict__ chain, double* out, Stamp* st, int iters, int stride_mask) {
  // latency of a dependent 4-byte load, every lane the same address (L1 / L2 / memory, by the footprint of the chain)
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  __syncthreads();
  int p = (blockIdx.x * 4099) & stride_mask;
  const long long r0 = wall_clock64();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p = __builtin_nontemporal_load(chain + p);
  }
  const long long t1 = clock64();
  const long long r1 = wall_clock64();
  if ((tid & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + tid / 64;
    st[w].cyc = t1 - t0;
    st[w].real = r1 - r0;
  }
  out[blockIdx.x * blockDim.x + tid] = p + lds[0] * 0.0;
}
template <int NB>
__global__ void __launch_bounds__(1024) k_gl_stream(const double* __restrict__ src, double* out, Stamp* st, int iters, int nelem_mask) {
  // throughput of 8-byte-per-lane coalesced loads (NB loads in flight per wave, then one wait): rows of 512 B per wave
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  __syncthreads();
  double acc = lds[0] * 0.0;
  unsigned base = (blockIdx.x * blockDim.x + tid) * 1u;
  const long long r0 = wall_clock64();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = src[(base + (unsigned)(it * NB + u) * 65536u) & (unsigned)nelem_mask];
#pragma unroll
    for (int u = 0; u < NB; ++u) acc += v[u];
  }
  const long long t1 = clock64();
  const long long r1 = wall_clock64();
  if ((tid & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + tid / 64;
    st[w].cyc = t1 - t0;
    st[w].real = r1 - r0;
  }
  out[blockIdx.x * blockDim.x + tid] = acc;
}
// the SHAPE of a wave-load: 8 bytes per lane, G consecutive lanes share a 128-byte line (G * 8 bytes of it), the wave touches
// 64 / G lines that are `line_stride` lines apart.  G = 16: four whole lines (k_gl_stream's shape); G = 4: sixteen lines for 32
// bytes each (the column solver's shape before its movers, the x-runs' operands of the transport kernels); G = 1: 64 lines.
// `share`: the waves of a workgroup ask for the SAME lines (different 8-byte pieces when G < 16 allows) -- the column solver's
// four waves -- or for lines of their own.
template <int NB, int G>
__global__ void __launch_bounds__(1024) k_gl_shape(const double* __restrict__ src, double* out, Stamp* st, int iters, int nelem_mask,
                                                   int line_stride) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  __syncthreads();
  double acc = lds[0] * 0.0;
  const int lane = tid & 63, wave = tid >> 6;
  // element offset of this lane inside the wave's footprint: line (lane / G) * line_stride, piece lane % G (8 bytes each)
  const unsigned in_wave = (unsigned)(lane / G) * 16u * (unsigned)line_stride + (unsigned)(lane % G);
  const unsigned foot = (64u / G) * 16u * (unsigned)line_stride;  // elements a wave-load spans
  unsigned base = (blockIdx.x * (blockDim.x >> 6) + wave) * foot + in_wave;
  const unsigned round = gridDim.x * (blockDim.x >> 6) * foot;  // elements all waves of the launch touch per load: the next load starts behind them
  const long long r0 = wall_clock64();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = src[(base + (unsigned)(it * NB + u) * round) & (unsigned)nelem_mask];
#pragma unroll
    for (int u = 0; u < NB; ++u) acc += v[u];
  }
  const long long t1 = clock64();
  const long long r1 = wall_clock64();
  if ((tid & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + tid / 64;
    st[w].cyc = t1 - t0;
    st[w].real = r1 - r0;
  }
  out[blockIdx.x * blockDim.x + tid] = acc;
}
// the same through LDS-DMA (16 B per lane, 1 KiB per wave-instruction), NB in flight then vmcnt(0)
template <int NB>
__global__ void __launch_bounds__(1024) k_gl_lds_dma(const double* __restrict__ src, double* out, Stamp* st, int iters, int nelem_mask) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  __syncthreads();
  unsigned base = (blockIdx.x * blockDim.x + tid) * 2u;  // doubles: 16 B per lane
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)lds + wave * (NB * 1024u);
  const long long r0 = wall_clock64();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const double* a = src + ((base + (unsigned)(it * NB + u) * 131072u) & (unsigned)nelem_mask);
      const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_base + u * 1024u);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(m0v) : "memory", "m0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  const long long r1 = wall_clock64();
  if ((tid & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + tid / 64;
    st[w].cyc = t1 - t0;
    st[w].real = r1 - r0;
  }
  out[blockIdx.x * blockDim.x + tid] = lds[tid];
}

struct Result {
  double cyc_per_instr_wave, mhz;
};

template <class F>
static Result run(F launch, int waves_per_simd, int iters, int instr_per_iter, Stamp* d_st, int nblocks) {
  const int threads = 256 * waves_per_simd;
  const int nw = nblocks * threads / 64;
  CHECK(hipMemset(d_st, 0, sizeof(Stamp) * nw));
  launch(nblocks, threads, iters);  // warm-up
  CHECK(hipDeviceSynchronize());
  launch(nblocks, threads, iters);
  CHECK(hipDeviceSynchronize());
  std::vector<Stamp> h(nw);
  CHECK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
  std::vector<double> c;
  double cyc = 0, real = 0;
  for (auto& s : h) {
    c.push_back((double)s.cyc);
    cyc += s.cyc;
    real += s.real;
  }
  std::sort(c.begin(), c.end());
  Result r;
  r.cyc_per_instr_wave = c[c.size() / 2] / ((double)iters * instr_per_iter);
  r.mhz = real > 0 ? cyc / real * 100.0 : 0.0;  // s_memrealtime ticks at 100 MHz
  return r;
}

int main(int argc, char** argv) {
  const int nblocks = 256;  // one per CU
  const size_t lds_bytes = 96 * 1024;
  double* d_seed;
  double* d_out;
  Stamp* d_st;
  CHECK(hipMalloc(&d_seed, 64));
  const double hseed[2] = {1.000000001, 3.0};
  CHECK(hipMemcpy(d_seed, hseed, 16, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_out, sizeof(double) * nblocks * 1024));
  CHECK(hipMalloc(&d_st, sizeof(Stamp) * nblocks * 16));
  const int iters = 2000;

#define VALU_TEST(name, kern, ipi)                                                                                      \
  do {                                                                                                                  \
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));          \
    for (int w : {1, 2, 4}) {                                                                                           \
      Result r = run([&](int nb, int th, int it) { hipLaunchKernelGGL(kern, dim3(nb), dim3(th), lds_bytes, 0, d_seed, d_out, d_st, it); }, w, iters, ipi, d_st, nblocks); \
      printf("%-22s waves/SIMD %d  cycles/instr seen by a wave %7.2f  per SIMD %6.2f   clock %6.0f MHz\n", name, w, r.cyc_per_instr_wave, \
             r.cyc_per_instr_wave / w, r.mhz);                                                                          \
    }                                                                                                                   \
  } while (0)

  VALU_TEST("f64_add_dependent", k_f64_add_dep, 8);
  VALU_TEST("f64_add_independent", k_f64_add_ind, 8);
  VALU_TEST("f64_mul_independent", k_f64_mul_ind, 8);
  VALU_TEST("f64_fma_independent", k_f64_fma_ind, 8);
  VALU_TEST("f64_rcp_independent", k_f64_rcp_ind, 8);
  VALU_TEST("f64_ieee_div (each)", k_f64_div, 8);
  VALU_TEST("i32_add_dependent", k_i32_add_dep, 8);
  VALU_TEST("i32_add_independent", k_i32_add_ind, 8);
  VALU_TEST("i32_mad24_independent", k_i32_mad24_ind, 8);
  VALU_TEST("cndmask_b32", k_cndmask_ind, 8);
  VALU_TEST("mix f64 + i32", k_mix_f64_i32, 8);
  VALU_TEST("mix f64 + salu", k_mix_valu_salu, 8);
  VALU_TEST("salu_add", k_salu_ind, 8);
  VALU_TEST("dpp wave_shr (+nop)", k_dpp_shr, 8);
  VALU_TEST("lds_read_b32 chase", k_lds_read_dep, 8);
  VALU_TEST("lds_read_b64 x8+wait", k_lds_read64_ind, 8);
  VALU_TEST("lds_write_b64 x8+wait", k_lds_write64_ind, 8);
  VALU_TEST("s_barrier", k_barrier, 8);

  // global memory: chase (latency) with footprints 16 KB (L1), 1 MB (L2), 64 MB (MALL), 1 GB (HBM)
  {
    const size_t nmax = (size_t)256 << 20;  // ints: 1 GB
    int* d_chain;
    CHECK(hipMalloc(&d_chain, nmax * 4));
    std::vector<int> h(nmax);
    for (size_t fp : {(size_t)4096, (size_t)262144, (size_t)16 << 20, nmax}) {
      // a stride-permuted cycle over the footprint, in units of 64 B lines
      const size_t nl = fp / 16;
      const size_t step = (nl > 16) ? (nl / 2 + 17) | 1 : 1;
      for (size_t l = 0; l < nl; ++l) {
        const size_t nxt = (l + step) % nl;
        for (int e = 0; e < 16; ++e) h[l * 16 + e] = (int)(nxt * 16 + e);
      }
      CHECK(hipMemcpy(d_chain, h.data(), fp * 4, hipMemcpyHostToDevice));
      CHECK(hipFuncSetAttribute((const void*)k_gl_chase, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      for (int w : {1}) {
        Result r = run([&](int nb, int th, int it) { hipLaunchKernelGGL(k_gl_chase, dim3(nb), dim3(th), lds_bytes, 0, d_chain, d_out, d_st, it, (int)(fp - 1)); }, w, 200, 8, d_st, nblocks);
        printf("global chase footprint %8zu KB   waves/SIMD %d  cycles/load %8.1f  clock %6.0f MHz\n", fp * 4 / 1024, w, r.cyc_per_instr_wave, r.mhz);
      }
    }
    CHECK(hipFree(d_chain));
  }
  // global streaming loads: 8 B / lane; buffer 16 MB (L2 / MALL resident after warm-up) and 2 GB (HBM)
  {
    const size_t nbig = (size_t)256 << 20;  // doubles: 2 GB
    double* d_src;
    CHECK(hipMalloc(&d_src, nbig * 8));
    CHECK(hipMemset(d_src, 0, nbig * 8));
    for (size_t ne : {(size_t)2 << 20, nbig}) {
#define STREAM_TEST(NB)                                                                                                    \
  do {                                                                                                                     \
    CHECK(hipFuncSetAttribute((const void*)k_gl_stream<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));  \
    for (int w : {1, 2, 4}) {                                                                                              \
      Result r = run([&](int nb, int th, int it) { hipLaunchKernelGGL(k_gl_stream<NB>, dim3(nb), dim3(th), lds_bytes, 0, d_src, d_out, d_st, it, (int)(ne - 1)); }, w, 400, NB, d_st, nblocks); \
      const double bpc = 512.0 * w * 4 / r.cyc_per_instr_wave;                                                             \
      printf("global 8B/lane loads, %d in flight, buffer %5zu MB  waves/SIMD %d  cycles/load seen by a wave %7.1f  B/clk/CU %6.1f  chip %5.2f TB/s  clock %6.0f MHz\n", \
             NB, ne * 8 >> 20, w, r.cyc_per_instr_wave, bpc, bpc * 256 * r.mhz * 1e6 / 1e12, r.mhz);                      \
    }                                                                                                                      \
  } while (0)
      STREAM_TEST(1);
      STREAM_TEST(4);
      STREAM_TEST(8);
#define DMA_TEST(NB)                                                                                                       \
  do {                                                                                                                     \
    CHECK(hipFuncSetAttribute((const void*)k_gl_lds_dma<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
    for (int w : {1, 2, 4}) {                                                                                              \
      Result r = run([&](int nb, int th, int it) { hipLaunchKernelGGL(k_gl_lds_dma<NB>, dim3(nb), dim3(th), lds_bytes, 0, d_src, d_out, d_st, it, (int)(ne - 1)); }, w, 400, NB, d_st, nblocks); \
      const double bpc = 1024.0 * w * 4 / r.cyc_per_instr_wave;                                                            \
      printf("LDS-DMA 16B/lane,     %d in flight, buffer %5zu MB  waves/SIMD %d  cycles/load seen by a wave %7.1f  B/clk/CU %6.1f  chip %5.2f TB/s  clock %6.0f MHz\n", \
             NB, ne * 8 >> 20, w, r.cyc_per_instr_wave, bpc, bpc * 256 * r.mhz * 1e6 / 1e12, r.mhz);                      \
    }                                                                                                                      \
  } while (0)
      DMA_TEST(1);
      DMA_TEST(4);
#define SHAPE_TEST(G, STRIDE)                                                                                              \
  do {                                                                                                                     \
    CHECK(hipFuncSetAttribute((const void*)k_gl_shape<4, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
    for (int w : {2, 4}) {                                                                                                 \
      Result r = run([&](int nb, int th, int it) { hipLaunchKernelGGL((k_gl_shape<4, G>), dim3(nb), dim3(th), lds_bytes, 0, d_src, d_out, d_st, it, (int)(ne - 1), STRIDE); }, w, 400, 4, d_st, nblocks); \
      const double bpc = 512.0 * w * 4 / r.cyc_per_instr_wave;                                                             \
      printf("shape: %2d lines x %3d B per wave-load, lines %3d apart, buffer %5zu MB  waves/SIMD %d  cycles/load seen by a wave %7.1f  useful B/clk/CU %6.1f  chip %5.2f TB/s useful\n", \
             64 / G, G * 8, STRIDE, ne * 8 >> 20, w, r.cyc_per_instr_wave, bpc, bpc * 256 * r.mhz * 1e6 / 1e12);          \
    }                                                                                                                      \
  } while (0)
      SHAPE_TEST(16, 1);
      SHAPE_TEST(8, 1);
      SHAPE_TEST(4, 1);
      SHAPE_TEST(2, 1);
      SHAPE_TEST(1, 1);
      SHAPE_TEST(4, 13);
      SHAPE_TEST(16, 13);
    }
    CHECK(hipFree(d_src));
  }
  return 0;
}
