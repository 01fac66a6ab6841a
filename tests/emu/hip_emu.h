/*
 * tests/emu/hip_emu.h -- host-side SIMT emulator for the kernel sources.  TEST TOOL ONLY.
 *
 * There is no GPU in the build container, and GPU minutes are scarce, so the kernel
 * sources under grayskull_amd/csrc/ can also be compiled with g++ -DGS_EMU against this
 * header.  It runs one workgroup at a time; every work-item is a ucontext fiber, scheduled
 * round-robin on ONE OS thread, so runs are deterministic and race-free by construction.
 * Barriers and wave-level exchanges (ballot / DPP shift / shuffles) yield between fibers.
 *
 * This checks INDEXING AND ARITHMETIC LOGIC of the kernels on tiny inputs.  It is not a
 * CPU fallback: the product library (libgrayskull_hip.so) never contains or loads it, and
 * the python package refuses to run without the HIP library.  What it cannot check
 * (hardware DPP/bounds semantics, unaligned vector access, occupancy, speed) is checked on
 * the GPU by the `-m gpu` tests.
 */
#ifndef HIP_EMU_H
#define HIP_EMU_H
#ifndef GS_EMU
#error "hip_emu.h is only for -DGS_EMU builds"
#endif

#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local /* one emulated device per host thread (gsbatch --gpus N) */
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx { unsigned x, y, z; };

typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };

namespace emu {
constexpr int WAVE = 64;
constexpr size_t STACK = 256 * 1024;

struct Fiber {
  ucontext_t ctx; /* first entry only; afterwards fibers switch through jb (no signal-mask system call per switch) */
  jmp_buf jb;
  bool started = false;
  char *stack = nullptr;
  bool done = true;
  emu_idx tid;
  unsigned lin = 0;
};

struct State {
  emu_idx bidx, tidx;
  dim3 bdim, gdim;
  std::vector<Fiber> fibers;
  ucontext_t sched;
  jmp_buf sched_jb;
  int cur = -1;
  unsigned nthreads = 0, alive = 0;
  /* block barrier */
  unsigned bar_count = 0, bar_gen = 0;
  /* per-wave exchange */
  struct WaveX {
    unsigned count = 0, gen = 0, alive = 0;
    uint64_t slot[WAVE];
    bool valid[WAVE];
    unsigned arrived[WAVE] = {}; /* generation + 1 of the rendezvous a lane is waiting in */
  };
  std::vector<WaveX> waves;
  /* per-quad exchange (DPP quad_perm): double-buffered slots, one rendezvous per exchange, and a lane that has
   * to wait hands the CPU straight to a sibling that has not arrived (not a full round of the block's fibers) */
  struct QuadX {
    unsigned count = 0, gen = 0;
    uint64_t slot[2][4];
  };
  std::vector<QuadX> quads;
  const std::function<void()> *body = nullptr;
  char *dyn_lds = nullptr;
  ~State() { /* one State per host thread: give the fiber stacks back when the thread ends */
    for (auto &f : fibers) free(f.stack);
  }
};
State &S();

void yield();
void block_barrier();
/* all live lanes of the calling wave publish `v`; returns after everyone has published.
 * `fn(slots, valid)` is evaluated by each lane before the closing rendezvous. */
template <class F> auto wave_exchange(uint64_t v, F fn) -> decltype(fn((const uint64_t *)0, (const bool *)0));
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
inline unsigned lane_id() { return S().fibers[S().cur].lin % WAVE; }
inline unsigned wave_id() { return S().fibers[S().cur].lin / WAVE; }
void wave_rendezvous();
/* the four lanes of the calling lane's quad swap `v`: returns the value published by lane (quad base + sel) */
uint64_t quad_exchange(uint64_t v, unsigned sel);
}  // namespace emu

#define threadIdx (emu::S().tidx)
#define blockIdx (emu::S().bidx)
#define blockDim (emu::S().bdim)
#define gridDim (emu::S().gdim)

inline void __syncthreads() { emu::block_barrier(); }

template <class F>
auto emu::wave_exchange(uint64_t v, F fn) -> decltype(fn((const uint64_t *)0, (const bool *)0)) {
  State &s = S();
  State::WaveX &w = s.waves[wave_id()];
  unsigned l = lane_id();
  w.slot[l] = v;
  w.valid[l] = true;
  wave_rendezvous(); /* everyone has written */
  auto r = fn(w.slot, w.valid);
  wave_rendezvous(); /* everyone has read */
  w.valid[l] = false;
  return r;
}

/* ---- launch macro shared with the HIP build ---- */
#define GS_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })

/* ---- atomics (single OS thread => plain ops are atomic) ---- */
template <class T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }

inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }

/* ---- tiny HIP runtime shim: "device" memory is host memory ---- */
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void *p) { free(p); return 0; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void *p) { free(p); return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t wbytes, size_t rows, int, hipStream_t) {
  for (size_t r = 0; r < rows; r++) memcpy((char *)d + r * dp, (const char *)s + r * sp, wbytes);
  return 0;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDeviceCount(int *n) { /* GS_EMU_DEVICES: pretend to have that many GPUs (multi-GPU host logic) */
  const char *e = getenv("GS_EMU_DEVICES");
  *n = e && atoi(e) > 0 ? atoi(e) : 1;
  return 0;
}
inline const char *hipGetErrorString(hipError_t) { return "emu"; }
#endif
