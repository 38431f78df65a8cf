// Fiber scheduler for tests/emu/hip_emu.h (test infrastructure only).
#include "hip_emu.h"

#include <mutex>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  bool done = false;
  uint3_emu tid;
};
ucontext_t g_main;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;

void trampoline() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_main);
}
}  // namespace

namespace {
bool g_plain = false;
struct NeedsFibers {};
}  // namespace

void __syncthreads() {
  if (g_plain) throw NeedsFibers();
  swapcontext(&g_cur->ctx, &g_main);
}

namespace {
unsigned char g_shfl[1024][16];
}

void emu_shuffle_exchange(const void* mine, void* out, int src_lane_in_wave, int nbytes) {
  if (g_plain) throw NeedsFibers();
  const unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  const unsigned wave_base = tid & ~63u;
  __builtin_memcpy(g_shfl[tid], mine, (size_t)nbytes);
  swapcontext(&g_cur->ctx, &g_main);  // every lane has deposited
  __builtin_memcpy(out, g_shfl[wave_base + (unsigned)src_lane_in_wave], (size_t)nbytes);
  swapcontext(&g_cur->ctx, &g_main);  // every lane has read
}

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body, int* mode) {
  // one launch at a time: the scheduler state is global (several tile threads may call into the library)
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  gridDim = grid;
  blockDim = block;
  g_body = &body;
  if (*mode != 2) {
    g_plain = true;
    try {
      for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
          for (unsigned bx = 0; bx < grid.x; ++bx) {
            blockIdx = {bx, by, bz};
            for (unsigned tz = 0; tz < block.z; ++tz)
              for (unsigned ty = 0; ty < block.y; ++ty)
                for (unsigned tx = 0; tx < block.x; ++tx) {
                  threadIdx = {tx, ty, tz};
                  body();
                }
          }
      g_plain = false;
      *mode = 1;
      return;
    } catch (const NeedsFibers&) {
      g_plain = false;
      *mode = 2;
    }
  }
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  const size_t stack_bytes = 256 * 1024;
  std::vector<Fiber> fibers(nthreads);
  for (auto& f : fibers) f.stack.resize(stack_bytes);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        size_t t = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
              Fiber& f = fibers[t];
              f.done = false;
              f.tid = {tx, ty, tz};
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = f.stack.data();
              f.ctx.uc_stack.ss_size = stack_bytes;
              f.ctx.uc_link = &g_main;
              makecontext(&f.ctx, trampoline, 0);
            }
        size_t remaining = nthreads;
        while (remaining) {
          for (auto& f : fibers) {
            if (f.done) continue;
            g_cur = &f;
            threadIdx = f.tid;
            blockIdx = {bx, by, bz};
            swapcontext(&g_main, &f.ctx);
            if (f.done) --remaining;
          }
        }
      }
}
