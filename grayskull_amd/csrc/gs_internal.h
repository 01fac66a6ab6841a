#ifndef GS_INTERNAL_H
#define GS_INTERNAL_H
/*
 * gs_internal.h -- what the translation units of libgrayskull_hip.so share: the per-thread context, the launch-tuning
 * table, the device topology, staging helpers and the launchers that cross unit boundaries.
 *   gs_ctx.cpp      context, device / stream / memory entry points of the C ABI
 *   gs_stencil.cpp  stencils, pointwise ops, integral, geometry, template matching (+ their C ABI)
 *   gs_detect.cpp   ordered compaction, FAST, LBP cascade, ORB, matching (+ their C ABI)
 *   gs_comm.cpp     RCCL control plane for one-process multi-GPU host programs
 *   gs_fused.cpp, gs_box.cpp  kernels with their own compiler flags
 *
 * libgrayskull_hip.so: host runtime and C-ABI.
 *
 * Exports (a) the reference's own function names/signatures for the hot path (declared in
 * include/grayskull.h, each citing the reference definition it replaces) and (b) the
 * device-resident batch entry points of include/grayskull_hip.h.  Every compute step is a
 * HIP kernel from k_*.h; the only arithmetic done on the host is what the reference itself
 * delegates to libm (atan2f / sinf, grayskull.h:100-101), the float32 scale progression of
 * gs_lbp_detect (ref :819-821, :799-804) and the stable sort of <= 5000 candidates (ref :639).
 * There is no CPU fallback: without a HIP device every entry point aborts.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>


#include "../../include/grayskull_hip.h"

#define GS_COMMA ,
#define GS_ASSERT(cond)                                 \
  do {                                                  \
    if (!(cond)) {                                      \
      fprintf(stderr, "Assertion failed: %s\n", #cond); \
      abort();                                          \
    }                                                   \
  } while (0)

#define GS_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "grayskull_hip: %s failed: %s (%s:%d)\n", #call,                 \
              hipGetErrorString(e_), __FILE__, __LINE__);                              \
      abort();                                                                         \
    }                                                                                  \
  } while (0)

#include "prims.h"
#include "lbp_types.h"
#include "k_strip.h" /* strip_ragged_shift / strip_realign_shift: host side of the lane placement */

namespace gs { /* gs_fused.cpp */
void launch_blur_sobel_hist(unsigned radius, dim3 grid, dim3 block, hipStream_t st, uint8_t *dst,
                            const uint8_t *src, unsigned w, unsigned h, unsigned T, size_t frame_bytes,
                            unsigned *partial);
void launch_blur_sobel(unsigned radius, dim3 grid, dim3 block, hipStream_t st, uint8_t *dst, const uint8_t *src,
                       unsigned w, unsigned h, unsigned T, size_t frame_bytes, int rg);
/* gs_box.cpp */
void launch_box(int mode, unsigned ring_radius, dim3 grid, unsigned threads, hipStream_t st, uint8_t *dst, const uint8_t *src, unsigned w,
                unsigned h, unsigned T, size_t frame_bytes, unsigned r, int c);
void launch_box_edge(int mode, dim3 grid, hipStream_t st, uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned T,
                     size_t frame_bytes, unsigned r, int c);
unsigned box_blocks_per_cu(int mode, unsigned ring_radius, unsigned threads);
unsigned box_ring_max();
}
using namespace gs;

namespace gsi {

/* ------------------------------------------------------------------ per-thread context */
enum Slot { SL_IN = 0, SL_OUT, SL_AUX, SL_AUX2, SL_II, SL_PAD, SL_MASK, SL_CNT, SL_PFX, SL_TOT, SL_PRE,
            SL_HISTP, SL_HIST, SL_THR, SL_KPS, SL_MOM, SL_KIN, SL_DESC, SL_TAB, SL_JUMP, SL_LEV,
            SL_BEST, SL_NZ, SL_COUNT };

/* gsh_edge_pipeline_batch: frames per chunk (measured best for 64..512-frame batches of 4K frames:
 * profiles/r01g_chunk_overlap.log) and the most chunks per call */
constexpr unsigned kChunkFrames = 32, kMaxChunks = 256; /* 8192 frames per call keep the chunk overlap */
/* per-(thread, cascade) scan geometry of the last gs_lbp_detect call: the scale list and the
 * per-(scale, classifier) corner offsets on the device.  Lives in the calling thread's context, not
 * in the cascade handle, so threads sharing one handle never touch each other's tables. */
struct LbpGeomCache {
  unsigned iw = 0, ih = 0;
  float sf = 0, mn = 0, mx = 0;
  int step = 0;
  std::vector<LbpScale> scales;
  LbpScale *d_scales = nullptr;
  LbpGeom *d_geom = nullptr;
  size_t d_scales_cap = 0, d_geom_cap = 0;
  unsigned total_chunks = 0, max_chunks = 0;
  bool guard = false;
  unsigned long long nwindows = 0;
};
struct gsh_cascade_tables_deleter { void operator()(struct ::gsh_cascade *dc) const; };
/* Events that order the library's own streams on ONE device need no system-scope fence: a plain
 * hipEventRecord releases to the system (L2 write-back of everything the previous kernel left dirty),
 * measured at ~37 us per record between back-to-back launches that each write 265 MB (scripts/ubench_gaps.py,
 * profiles/r02f_launch_gaps.log; inside the pipeline the side stream's threshold pass fills that gap, so
 * the step time moves by < 0.5 %, profiles/r02f_event_flags.log).  GS_EVENT_FLAGS (experiment hook): 0 = HIP's default. */
#ifndef GS_EMU
#ifndef GS_EVENT_FLAGS
#define GS_EVENT_FLAGS hipEventDisableSystemFence
#endif
inline unsigned sync_event_flags() { return GS_EVENT_FLAGS; } /* timing-only events (gsh_profile) */
/* The events that ORDER the side stream against the caller's stream (ev_join / ev_chunk) carry the data
 * dependence of dst / thr / the partial histograms between the two streams, so they keep HIP's documented
 * semantics (a release the waiting stream is guaranteed to observe) -- hipEventDisableSystemFence is documented
 * for timing-only events and worked for ordering only through ROCclr's kernel-boundary release.
 * GS_ORDER_EVENT_FLAGS (experiment hook) adds flags to them. */
#ifndef GS_ORDER_EVENT_FLAGS
#define GS_ORDER_EVENT_FLAGS 0
#endif
inline unsigned order_event_flags() { return hipEventDisableTiming | GS_ORDER_EVENT_FLAGS; }
#endif
/* What the launch heuristics need to know about the device, read once per context from the runtime instead of MI355X
 * literals (round 4): a CPX-partitioned or smaller part reports fewer CUs / one XCD, and then the band counts scale
 * with it and the XCD-aware block mappings (which assume the dispatcher's round robin over EIGHT dies) stay off. */
struct Topo {
  unsigned cus = 256, xcds = 8;
  unsigned simds() const { return cus * 4u; } /* CDNA: four SIMDs per CU */
  bool eight_xcds() const { return xcds == 8u; }
};
struct Ctx {
  int device = 0;
  bool device_set = false;
  Topo topo;
  hipStream_t stream = nullptr;
  bool own_stream = false, user_stream = false, async = false;
  struct Buf { void *p = nullptr; size_t cap = 0; } slot[SL_COUNT];
  /* grow-only PINNED host staging (hipHostMalloc) for the few paths that bring records back to the host inside a call: a
   * copy into pageable memory is staged by the runtime at a few GB/s and blocks the stream meanwhile */
  enum { PIN_A = 0, PIN_B, PIN_C, PIN_D, PIN_COUNT };
  Buf pin[PIN_COUNT];
  void *pinned(int i, size_t bytes) {
    ensure_device();
    Buf &b = pin[i];
    if (b.cap < bytes) {
      if (b.p) {
        sync();
        GS_HIP(hipHostFree(b.p));
      }
      const size_t cap = bytes + bytes / 4 + 256;
      GS_HIP(hipHostMalloc(&b.p, cap, 0));
      b.cap = cap;
    }
    return b.p;
  }
  bool jump_ready = false; /* SL_JUMP holds the xorshift jump table of gsh_synth_batch */
  bool lbp_lds_raised = false; /* k_lbp_tile's hipFuncAttributeMaxDynamicSharedMemorySize set on this device (gs_detect.cpp) */
  std::map<unsigned long long, LbpGeomCache> geom_cache; /* keyed by gsh_cascade::id */
  /* flattened copy of the caller's struct gs_lbp_cascade for the drop-in gs_lbp_* calls, keyed by a hash
   * of the table contents (cached_cascade); owned here so that it goes away with the context, in order */
  struct ::gsh_cascade *dropin_cascade = nullptr;
  uint64_t dropin_cascade_hash = 0;
  void drop_geom() {
    for (auto &kv : geom_cache) {
      if (kv.second.d_scales) (void)hipFree(kv.second.d_scales);
      if (kv.second.d_geom) (void)hipFree(kv.second.d_geom);
    }
    geom_cache.clear();
  }
  /* a host thread that drives one GPU and exits (gsbatch --gpus N) gives its scratch, streams and
   * events back without having to remember gsh_shutdown() */
  ~Ctx() { release(); }
#ifndef GS_EMU
  /* gsh_profile: events bracketing the pipeline's fused-kernel launches on their stream */
  static constexpr int kProfPairs = 4096;
  bool prof_on = false;
  hipEvent_t prof_ev[2 * kProfPairs] = {};
  unsigned prof_n = 0;
  void prof_mark(int which, hipStream_t on) { /* which: 0 before, 1 after the launch */
    if (!prof_on || prof_n >= (unsigned)kProfPairs) return;
    hipEvent_t &e = prof_ev[2 * prof_n + which];
    if (!e) GS_HIP(hipEventCreateWithFlags(&e, sync_event_flags())); /* normally pre-created by gsh_profile */
    GS_HIP(hipEventRecord(e, on));
    if (which) prof_n++;
  }
  hipStream_t side = nullptr; /* chunk overlap inside gsh_edge_pipeline_batch */
  hipEvent_t ev_join = nullptr, ev_chunk[kMaxChunks] = {};
  void ensure_side() {
    if (side) return;
    {
      /* A stream of default priority can land on the same hardware queue as the caller's stream
       * (HIP hands its queues out round-robin; measured: the second such stream created in a
       * process serialised behind the main stream, 2.1 ms instead of 1.65 ms per 256-frame step).
       * Streams of another priority level have their own queues, so take the lowest priority --
       * the passes on this stream should yield to the fused kernels anyway -- unless the caller's
       * stream is itself of that priority. */
      int lo = 0, hi = 0, mine = 0;
      GS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi)); /* numerically lower = higher priority */
      if (hipStreamGetPriority(s(), &mine) != hipSuccess) mine = 0;
      GS_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, mine == lo && lo != hi ? hi : lo));
    }
    GS_HIP(hipEventCreateWithFlags(&ev_join, order_event_flags()));
    for (auto &e : ev_chunk) GS_HIP(hipEventCreateWithFlags(&e, order_event_flags()));
  }
#endif

  void ensure_device() {
    if (device_set) return;
    int n = 0;
#ifndef GS_EMU
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
      fprintf(stderr, "grayskull_hip: no HIP device visible (%s); there is no CPU fallback\n",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
      abort();
    }
#endif
    GS_HIP(hipSetDevice(device));
    device_set = true;
#ifndef GS_EMU
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) topo.cus = (unsigned)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && v > 0) topo.xcds = (unsigned)v;
    else (void)hipGetLastError();
#endif
  }
  hipStream_t s() {
    ensure_device();
    if (user_stream) return stream;
    if (!own_stream) {
      GS_HIP(hipStreamCreate(&stream));
      own_stream = true;
    }
    return stream;
  }
  void sync() { GS_HIP(hipStreamSynchronize(s())); }
  /* grow-only device scratch; growing synchronises (old buffer may be in flight) */
  void *scratch(int i, size_t bytes) {
    ensure_device();
    Buf &b = slot[i];
    if (b.cap < bytes) {
      if (b.p) {
        sync();
        GS_HIP(hipFree(b.p));
      }
      size_t cap = bytes + bytes / 4 + 256;
      GS_HIP(hipMalloc(&b.p, cap));
      b.cap = cap;
    }
    return b.p;
  }
  void release() {
    for (auto &b : slot) {
      if (b.p) (void)hipFree(b.p);
      b.p = nullptr, b.cap = 0;
    }
    for (auto &b : pin) {
      if (b.p) (void)hipHostFree(b.p);
      b.p = nullptr, b.cap = 0;
    }
    jump_ready = false;
    lbp_lds_raised = false;
    drop_geom();
    if (dropin_cascade) gsh_cascade_tables_deleter()(dropin_cascade);
    dropin_cascade = nullptr;
#ifndef GS_EMU
    for (auto &e : prof_ev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    prof_n = 0;
    if (side) {
      (void)hipStreamDestroy(side), (void)hipEventDestroy(ev_join);
      for (auto &e : ev_chunk) (void)hipEventDestroy(e), e = nullptr;
      side = nullptr, ev_join = nullptr;
    }
#endif
    if (own_stream) (void)hipStreamDestroy(stream);
    own_stream = false;
    if (!user_stream) stream = nullptr;
  }
};
Ctx &ctx(); /* gs_ctx.cpp: one per calling thread */

inline bool is_dev(const void *p) {
#ifdef GS_EMU
  (void)p;
  return false;
#else
  if (!p) return false;
  ctx().ensure_device();
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); /* plain host memory: not an error for us */
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
#endif
}

/* host buffer -> device scratch (or pass a device pointer through) */
inline const void *stage_in(const void *p, size_t bytes, int slot) {
  if (is_dev(p)) return p;
  void *d = ctx().scratch(slot, bytes);
  GS_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx().s()));
  return d;
}
inline void finish(bool any_host_output) {
  if (any_host_output || !ctx().async) ctx().sync();
}
inline dim3 grid2d(unsigned w, unsigned h, unsigned n) { return dim3((w + 63) / 64, (h + 3) / 4, n); }
inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

/* ---- launch tuning (gsh_tune): 0 rows per band (0 = auto), 1 block shape, ...  Experiment / test hooks with PROCESS-WIDE
 * scope by design (a measurement script flips a key and every thread's next launch sees it); each entry is an atomic,
 * so concurrent host threads (gsbatch --gpus N workers) read whole values.  Nothing in a product path writes them. */
struct TuneTable {
  std::atomic<int> v[32];
  TuneTable() {
    for (auto &e : v) e.store(0, std::memory_order_relaxed);
    v[1].store(3, std::memory_order_relaxed), v[2].store(1, std::memory_order_relaxed);
  }
  int operator[](int k) const { return v[k].load(std::memory_order_relaxed); }
  void set(int k, int x) { v[k].store(x, std::memory_order_relaxed); }
};
extern TuneTable g_tune; /* gs_ctx.cpp */
inline const Topo &topo() {
  ctx().ensure_device();
  return ctx().topo;
}/* gsh_lbp_count_evaluated: device counter that receives the windows the cascade really evaluated (gs_detect.cpp) */
extern thread_local unsigned long long *g_lbp_evaluated;

struct StripCfg {
  dim3 grid, block;
  unsigned T;
  size_t xcd_flag = 0; /* OR into the kernel's frame_bytes argument: XCD-aware band mapping (k_strip.h) */
};/* gs_stencil.cpp (the comment there has the measurements behind the band heights) */
StripCfg strip_cfg(unsigned w, unsigned rows, unsigned n, unsigned waves_per_simd = 5, unsigned halo_rows = 2,
                   unsigned short_T = 8, int rg = -1);
/* The strip kernels take any width >= 32 and any byte alignment of the frames (round 4): rows of a frame whose width
 * is not a multiple of 16 start at every 16-byte phase anyway (the API has no stride, ref grayskull.h:14-17), the
 * hardware serves 16-byte accesses at any address, and the ragged last strip of a row is anchored at w - 16 (k_strip.h,
 * RAGGED).  Key 21 = 1 restores the round-3 rule (multiples of 16, 16-byte aligned frames; everything else per pixel). */
inline bool strip_ok(unsigned w, unsigned h, const void *a, const void *b) {
  if (g_tune[21] == 1) return w % 16 == 0 && (unsigned long long)w * h < 0x7fffffffull && al16(a) && al16(b);
  return w >= 32 && (unsigned long long)w * h < 0x7fff0000ull; /* a row offset + base phase + column must not wrap 2^32 */
}
inline bool ragged(unsigned w) { return (w & 15u) != 0u; }
/* which Strip flavour (k_strip.h): 0 = whole 16-px strips, rows at dword-aligned addresses; 1 = ragged width, rows still at
 * dword-aligned addresses (w % 4 == 0 and the frame at such an address): direct 16-byte loads at any 16-byte phase cost
 * nothing; 2 = any other byte phase: dword-aligned loads, realigned in registers (a 16-byte load at an address that is not
 * a multiple of 4 costs the stencils 30-45 %, profiles/r04b_byte_phase_cost.log).  Key 24 = 1: never 2 (A/B). */
inline int strip_mode(unsigned w, const void *src) {
  const int direct = ragged(w) ? 1 : 0;
  if ((w & 3u) == 0u && ((uintptr_t)src & 3u) == 0u) return direct;
  if (g_tune[24] == 1) return direct;
  /* the realigning flavour places up to two more lanes per row (k_strip.h); where that opens another wave of 64 -- widths
   * just below a multiple of 1024 -- the wave costs more than the misaligned loads do (4094 x 4096: 0.39 against 0.64 of
   * the HBM peak, profiles/r04g_ragged_realign.log) */
  const unsigned strips = (w + 15) / 16;
  const unsigned lanes2 = strips + strip_realign_shift(w) + strip_realign_helper(w), lanes1 = strips + (direct ? strip_ragged_shift(w) : 0u);
  if (g_tune[24] != 2 && (lanes2 + 63) / 64 > (lanes1 + 63) / 64) return direct;
  return 2;
}
/* kernels that still need whole 16-px strips at 16-byte aligned addresses */
inline bool strip_ok16(unsigned w, unsigned h, const void *a, const void *b) {
  return w % 16 == 0 && (unsigned long long)w * h < 0x7fffffffull && al16(a) && al16(b);
}
/* frames per launch (grid.y / grid.z limit 65535); gsh_tune key 8 lowers it so that the splitting
 * logic of every launcher can be exercised with a handful of frames */
inline unsigned max_frames_per_launch() { return g_tune[8] > 0 ? (unsigned)g_tune[8] : 32768u; }
#define kMaxZ (max_frames_per_launch())

#define GS_VALID(i) ((i).data && (i).w > 0 && (i).h > 0)

/* launchers that cross translation units */
void launch_sobel(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n, bool keep_cols = true); /* gs_stencil.cpp */
void launch_blur(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n, unsigned radius);
bool launch_integral(const uint8_t *src, unsigned w, unsigned h, unsigned n, unsigned *ii, bool sq = false);
void launch_integral_pad(dim3 grid, hipStream_t st, const unsigned *ii, unsigned w, unsigned h, unsigned *padded); /* k_integral_pad */

}  // namespace gsi
using namespace gs;
using namespace gsi;
#endif
