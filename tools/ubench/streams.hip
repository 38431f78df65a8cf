// Development tool (not product): the ceiling of the chip's memory path for THIS repo's access shapes -- read-only, write-only and
// copy kernels at 8 and 16 bytes per lane, with and without the nontemporal hint, over buffers far larger than the 256 MB
// Infinity Cache, launched as a plain grid (one element per thread) and as a persistent grid (8 workgroups per CU, grid-stride).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/streams.hip -o build/ubench_streams ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct alignas(16) D2 { double x, y; };

template <class T> __device__ __forceinline__ T ld(const T* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <class T> __device__ __forceinline__ void st(T* p, T v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ D2 ld2(const D2* p, bool nt) {
  D2 r;
  if (nt) { r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); } else r = *p;
  return r;
}
__device__ __forceinline__ void st2(D2* p, D2 v, bool nt) {
  if (nt) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); } else *p = v;
}

// MODE 0 read (sum kept live through a never-taken store), 1 write, 2 copy; W = 8 or 16 bytes per lane
template <int MODE, int W, bool NT>
__global__ void __launch_bounds__(256) k_stream(const double* __restrict__ src, double* __restrict__ dst, long n, double* sink) {
  const long stride = (long)gridDim.x * blockDim.x;
  double acc = 0.0;
  if (W == 8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      if (MODE == 0) acc += ld(src + i, NT);
      if (MODE == 1) st(dst + i, 1.0, NT);
      if (MODE == 2) st(dst + i, ld(src + i, NT), NT);
    }
  } else {
    const D2* s2 = (const D2*)src;
    D2* d2 = (D2*)dst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += stride) {
      if (MODE == 0) { D2 v = ld2(s2 + i, NT); acc += v.x + v.y; }
      if (MODE == 1) st2(d2 + i, D2{1.0, 2.0}, NT);
      if (MODE == 2) st2(d2 + i, ld2(s2 + i, NT), NT);
    }
  }
  if (MODE == 0 && acc == 1.2345e300) *sink = acc;
}

template <int MODE, int W, bool NT>
double run(const double* src, double* dst, long n, double* sink, bool persistent, int cus) {
  const long items = W == 8 ? n : n / 2;
  const unsigned grid = persistent ? 8u * cus : (unsigned)((items + 255) / 256);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ms;
  for (int rep = 0; rep < 7; ++rep) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_stream<MODE, W, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float t; hipEventElapsedTime(&t, a, b);
    if (rep >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (MODE == 2 ? 2.0 : 1.0) * n * 8.0;
  return bytes / (ms[ms.size() / 2] * 1e-3) / 1e12;  // TB/s, median
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const long n = 1L << 27;  // 1 GiB of doubles per buffer
  double *src, *dst, *sink;
  hipMalloc(&src, n * 8); hipMalloc(&dst, n * 8); hipMalloc(&sink, 8);
  hipMemset(src, 0, n * 8); hipMemset(dst, 0, n * 8);
  printf("%s, %d CUs; 1 GiB per buffer; TB/s (read and write bytes both counted for copy), median of 5\n", p.name, cus);
  printf("%-28s %10s %10s\n", "kernel", "plain grid", "persistent");
#define ROW(name, M, W, NT) printf("%-28s %10.2f %10.2f\n", name, run<M, W, NT>(src, dst, n, sink, false, cus), run<M, W, NT>(src, dst, n, sink, true, cus));
  ROW("read   8 B/lane", 0, 8, false) ROW("read   8 B/lane nt", 0, 8, true)
  ROW("read  16 B/lane", 0, 16, false) ROW("read  16 B/lane nt", 0, 16, true)
  ROW("write  8 B/lane", 1, 8, false) ROW("write  8 B/lane nt", 1, 8, true)
  ROW("write 16 B/lane", 1, 16, false) ROW("write 16 B/lane nt", 1, 16, true)
  ROW("copy   8 B/lane", 2, 8, false) ROW("copy   8 B/lane nt", 2, 8, true)
  ROW("copy  16 B/lane", 2, 16, false) ROW("copy  16 B/lane nt", 2, 16, true)
  // hipMemcpyDtoD for reference
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) { hipEventRecord(a, 0); hipMemcpyAsync(dst, src, n * 8, hipMemcpyDeviceToDevice, 0); hipEventRecord(b, 0); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); best = std::min(best, t); }
  printf("%-28s %10.2f\n", "hipMemcpyDtoD (best of 5)", 2.0 * n * 8 / (best * 1e-3) / 1e12);
  return 0;
}
