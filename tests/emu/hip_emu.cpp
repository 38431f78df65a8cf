// Fiber scheduler for tests/emu/hip_emu.h (test infrastructure only).
#include "hip_emu.h"

#include <mutex>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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

namespace {
const char* volatile g_kernel_name = "(no kernel)";
char g_altstack[64 * 1024];

// async-signal-safe enough for a test tool: format by hand, write(2), _exit
void put(const char* s) { (void)!write(2, s, strlen(s)); }
void put_num(unsigned long v, int base) {
  char buf[32];
  int n = 0;
  do {
    const int d = (int)(v % (unsigned)base);
    buf[n++] = (char)(d < 10 ? '0' + d : 'a' + d - 10);
    v /= (unsigned)base;
  } while (v);
  while (n) (void)!write(2, &buf[--n], 1);
}
void on_segv(int, siginfo_t* si, void*) {
  put("[emu] invalid access at 0x");
  put_num((unsigned long)si->si_addr, 16);
  put(" in kernel ");
  put(g_kernel_name);
  put(" block (");
  put_num(blockIdx.x, 10), put(","), put_num(blockIdx.y, 10), put(","), put_num(blockIdx.z, 10);
  put(") thread (");
  put_num(threadIdx.x, 10), put(","), put_num(threadIdx.y, 10), put(","), put_num(threadIdx.z, 10);
  put(")\n");
  _exit(99);
}
struct GuardInit {
  GuardInit() {
    if (!getenv("PACE_EMU_GUARD")) return;
    stack_t ss{};
    ss.ss_sp = g_altstack;
    ss.ss_size = sizeof(g_altstack);
    sigaltstack(&ss, nullptr);  // (per thread: the reporter is reliable for kernels launched from the loading thread)
    struct sigaction sa {};
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
  }
} g_guard_init;
}  // namespace

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body, int* mode, const char* name) {
  // one launch at a time: the scheduler state is global (several tile threads may call into the library)
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  g_kernel_name = name;
  gridDim = grid;
  blockDim = block;
  g_body = &body;
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  const size_t stack_bytes = 256 * 1024;
  std::vector<Fiber> fibers;
  // Block by block: the threads run as plain calls until one of them reaches a barrier, then THAT block is run again with
  // fibers (nothing but LDS is written before a block's first barrier: its first thread is the one that gets there).  A launch
  // may mix workgroups that synchronise with workgroups that do not (k_divdamp_fused: point-function workgroups first, then the
  // tiles) -- restarting the whole launch would run the former twice, read-modify-writes included.
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        bool need_fibers = *mode == 2;
        if (!need_fibers) {
          g_plain = true;
          try {
            for (unsigned tz = 0; tz < block.z; ++tz)
              for (unsigned ty = 0; ty < block.y; ++ty)
                for (unsigned tx = 0; tx < block.x; ++tx) {
                  threadIdx = {tx, ty, tz};
                  body();
                }
          } catch (const NeedsFibers&) {
            need_fibers = true;
          }
          g_plain = false;
        }
        if (!need_fibers) continue;
        if (fibers.empty()) {
          fibers.resize(nthreads);
          for (auto& f : fibers) f.stack.resize(stack_bytes);
        }
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
  (void)mode;
}
