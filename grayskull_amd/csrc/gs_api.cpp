/*
 * gs_api.cpp -- host runtime and C-ABI of libgrayskull_hip.so.
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
#include <map>
#include <vector>

#include "k_fast.h"
#include "k_fast_fused.h"
#include "k_fast_nms.h"
#include "k_geom.h"
#include "k_integral.h"
#include "k_lbp.h"
#include "k_lbp_dense.h"
#include "k_orb.h"
#include "k_pointwise.h"
#include "k_stencil.h"
#include "k_tmatch.h"

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

namespace gs { /* gs_fused.cpp */
void launch_blur_sobel_hist(unsigned radius, dim3 grid, dim3 block, hipStream_t st, uint8_t *dst,
                            const uint8_t *src, unsigned w, unsigned h, unsigned T, size_t frame_bytes,
                            unsigned *partial);
void launch_blur_sobel(unsigned radius, dim3 grid, dim3 block, hipStream_t st, uint8_t *dst, const uint8_t *src,
                       unsigned w, unsigned h, unsigned T, size_t frame_bytes, int rg);
/* gs_box.cpp */
void launch_box(int mode, unsigned ring_radius, dim3 grid, unsigned threads, hipStream_t st, uint8_t *dst, const uint8_t *src, unsigned w,
                unsigned h, unsigned T, size_t frame_bytes, unsigned r, int c);
unsigned box_blocks_per_cu(int mode, unsigned ring_radius, unsigned threads);
unsigned box_ring_max();
}
using namespace gs;

namespace {

/* ------------------------------------------------------------------ per-thread context */
enum Slot { SL_IN = 0, SL_OUT, SL_AUX, SL_AUX2, SL_II, SL_PAD, SL_MASK, SL_CNT, SL_PFX, SL_TOT, SL_PRE, SL_NOTII,
            SL_HISTP, SL_HIST, SL_THR, SL_KPS, SL_MOM, SL_KIN, SL_DESC, SL_TAB, SL_JUMP, SL_LEV,
            SL_BEST, SL_COUNT };

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
  /* prefilter geometry (k_lbp_dense.h), only built for step == 1 */
  LbpPreScale *d_pre = nullptr;
  size_t d_pre_cap = 0;
  unsigned long long pre_words = 0; /* u64 words of one frame's "alive" bitmap */
  unsigned long long max_cell_px = 0; /* largest fw x fh over all (scale, classifier) */
  unsigned max_tiles = 0;
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
  bool jump_ready = false; /* SL_JUMP holds the xorshift jump table of gsh_synth_batch */
  std::map<unsigned long long, LbpGeomCache> geom_cache; /* keyed by gsh_cascade::id */
  /* flattened copy of the caller's struct gs_lbp_cascade for the drop-in gs_lbp_* calls, keyed by a hash
   * of the table contents (cached_cascade); owned here so that it goes away with the context, in order */
  struct ::gsh_cascade *dropin_cascade = nullptr;
  uint64_t dropin_cascade_hash = 0;
  void drop_geom() {
    for (auto &kv : geom_cache) {
      if (kv.second.d_scales) (void)hipFree(kv.second.d_scales);
      if (kv.second.d_geom) (void)hipFree(kv.second.d_geom);
      if (kv.second.d_pre) (void)hipFree(kv.second.d_pre);
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
    jump_ready = false;
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
Ctx &ctx() {
  static thread_local Ctx c;
  return c;
}

bool is_dev(const void *p) {
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
const void *stage_in(const void *p, size_t bytes, int slot) {
  if (is_dev(p)) return p;
  void *d = ctx().scratch(slot, bytes);
  GS_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx().s()));
  return d;
}
void finish(bool any_host_output) {
  if (any_host_output || !ctx().async) ctx().sync();
}
dim3 grid2d(unsigned w, unsigned h, unsigned n) { return dim3((w + 63) / 64, (h + 3) / 4, n); }
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
TuneTable g_tune;
inline const Topo &topo() {
  ctx().ensure_device();
  return ctx().topo;
}
/* gsh_lbp_count_evaluated: device counter that receives the windows the cascade really evaluated */
thread_local unsigned long long *g_lbp_evaluated = nullptr;

struct StripCfg {
  dim3 grid, block;
  unsigned T;
  size_t xcd_flag = 0; /* OR into the kernel's frame_bytes argument: XCD-aware band mapping (k_strip.h) */
};
/* rows: output rows per frame.  One wave per (1024-px column block, band, frame).
 * HBM-bound per-call kernels (waves_per_simd >= 5) take SHORT bands of 8 rows (more for wide halos):
 * blocks are dispatched in band order, so the few thousand waves that are resident at any time work
 * on neighbouring rows of a few frames -- the DRAM pages they stream through are shared and the
 * halo rows of the next band are still in the Infinity Cache -- instead of each wave streaming its
 * own distant band.  Measured (profiles/r02g_small_bands_*.log, 64 frames): gs_sobel 4096x4096
 * 4.87 -> 5.62 TB/s (0.61 -> 0.70 of 8 TB/s) although it re-reads 2 halo rows per 8, 3840x2160
 * 4.76 -> 5.23; gs_blur(2) 4.88 -> 5.34 / 4.65 -> 4.90; erode 5.05 -> 5.57; plain strip copy 5.08 ->
 * 5.59.  Round 1 sized bands so that every wave of the launch was resident at once (20 bands of
 * 108 rows for 64 4K frames) and never tried bands below 16 rows.
 * The VALU-heavy fused kernels (waves_per_simd 3) keep long bands: each band first recomputes
 * 2R+2 rows of horizontal sums (8-row bands: 0.25 -> 0.38 ms per 64 frames). */
StripCfg strip_cfg(unsigned w, unsigned rows, unsigned n, unsigned waves_per_simd = 5, unsigned halo_rows = 2,
                   unsigned short_T = 8, int rg = -1) {
  StripCfg c;
  /* lanes to place (k_strip.h): ragged rows may idle lane 0, the realigning flavour uses the lane behind the last one.
   * rg = -1: the caller's kernel has no flavours (or ignores the extra lane) */
  const unsigned strips = (w + 15) / 16 + (rg == 2 ? strip_realign_shift(w) + strip_realign_helper(w) : rg == 1 ? strip_ragged_shift(w) : (w & 15u) ? 1u : 0u);
  const unsigned long long waves_x = (strips + 63) / 64;
  unsigned long long t;
  if (g_tune[0] > 0) {
    t = (unsigned long long)g_tune[0];
  } else if (waves_per_simd >= 5) {
    /* round 3, with the XCD-aware band mapping (a band's halo rows were just fetched by its own XCD) even shorter bands
     * pay: gs_sobel 6 rows (4096^2 0.706 -> 0.720 of the peak, 512 x 4K 0.705 -> 0.722; odd heights break its 2-row
     * unroll groups), gs_blur(2) 6 (0.671 -> 0.703 / 0.692 -> 0.705), gs_blur(1) and the morphology 4 (0.707 -> 0.730,
     * 0.716 -> 0.738 / 0.728 -> 0.746); profiles/r03l_strip_band_height_xcd.log.  short_T is the caller's choice. */
    t = std::max(short_T, 2u * halo_rows);
  } else {
    /* bands per frame: the launch should fill the chip's resident-wave capacity a whole number of
     * times.  One round when the batch is small enough; for big batches (512 4K frames: 2048 wave
     * columns against 3072 slots at 3 waves per SIMD) one band per frame would leave a third of the
     * chip idle for the whole launch and two bands would run a 1.33rd round at a third occupancy, so
     * take the band count whose last round is fullest (here 3: exactly two rounds; gsh_blur_sobel_batch
     * 512 frames 1.96 -> 1.89 ms, 200 frames 0.90 -> 0.74 ms, profiles/r02f_band_count_512.log). */
    const unsigned long long cap = (unsigned long long)topo().simds() * waves_per_simd, wn = waves_x * n;
    const unsigned long long nb_max = rows / 8 ? rows / 8 : 1; /* bands of >= 8 rows */
    unsigned long long nb = cap / wn;
    if (nb < 1) nb = 1;
    if (nb > nb_max) nb = nb_max;
    /* (only for the VALU-heavy fused kernels, which run 3 waves per SIMD: the HBM-bound per-call
     * kernels lose 3-6 % to the extra halo rows of shorter bands, profiles/r02f_band_count_512.log) */
    if (waves_per_simd <= 3 && nb < 4 && wn * nb * 8 < cap * 7) { /* a lone round under 7/8 full: try 2..6 rounds */
      double best = (double)((wn * nb + cap - 1) / cap) * cap / (double)(wn * nb);
      for (unsigned long long c = nb + 1; c <= nb_max && c <= nb + 6; c++) {
        const double waste = (double)((wn * c + cap - 1) / cap) * cap / (double)(wn * c);
        if (waste < best - 0.02) best = waste, nb = c;
      }
    }
    t = (rows + nb - 1) / nb;
  }
  c.T = (unsigned)t;
  const unsigned nb = (rows + c.T - 1) / c.T;
  /* block shape: 256 threads = 4 waves; a wave is 1024 px of a row.  Frames narrower than 4096 px put the spare
   * waves on further BANDS (64 x 4 up to 1024 px, 128 x 2 up to 2048 px) instead of columns that do not exist --
   * a 1920-px row kept two of a 256 x 1 block's four waves busy computing on zero fill (8 x 1080p gs_sobel at 0.21
   * of the HBM peak).  Key 1: 0 / 1 / 2 force 64 x 4 / 256 x 1 / 128 x 2, anything else = by width. */
  unsigned bx = 256, by = 1;
  if (g_tune[1] == 0 || (g_tune[1] > 2 && strips <= 64)) bx = 64, by = 4;
  else if (g_tune[1] == 2 || (g_tune[1] > 2 && strips <= 128)) bx = 128, by = 2;
  c.block = dim3(bx, by);
  c.grid = dim3((strips + bx - 1) / bx, (nb + by - 1) / by, n);
  /* XCD-aware band mapping for the short-band (HBM-bound) kernels: one block per band row (w <= 4096), the band
   * count padded to a multiple of 8 (blocks past the last band return at once).  Key 18: 1 = off, 2 = always. */
  /* measured (profiles/r03g_strip_xcd_bands.log): 512 x 4K gs_blur(2) +3.1 %, gs_erode +2.6 %, gs_sobel +2.0 %,
   * 64 x 4096^2 copy +2 %, sobel -1 % (noise); gs_filter (waves_per_simd 6) -3.5 %: not for that one */
  const bool want = g_tune[18] == 2 || (g_tune[18] == 0 && waves_per_simd == 5 && nb >= 64 && topo().eight_xcds());
  if (want && c.grid.x == 1 && by == 1) {
    c.grid.y = (nb + 7u) & ~7u;
    c.xcd_flag = kStripXcdFlag;
  }
  return c;
}
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

/* ------------------------------------------------------------------ stencil launchers */
/* keep_cols: true = columns 0 / w-1 keep dst's bytes like the reference (the kernel re-writes
 * them unchanged); false = the caller does not care (it copies back the interior only, or zeroes
 * the frame afterwards), which saves one dword load per row in the two edge lanes. */
void launch_sobel(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                  bool keep_cols = true) {
  if (w < 3 || h < 3 || n == 0) return;
  if (g_tune[22] == 1) keep_cols = false; /* probe: without the dst column reads (columns 0 / w-1 then receive junk) */
  hipStream_t st = ctx().s();
  const size_t fb = (size_t)w * h;
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    uint8_t *d = dst + fb * f0;
    const uint8_t *s = src + fb * f0;
    if (strip_ok(w, h, d, s) && w >= 32) {
      const int rg = strip_mode(w, s);
      const StripCfg c = strip_cfg(w, h - 2, nn, 5, 2, 6, rg);
      if (rg == 1) {
        if (keep_cols) GS_LAUNCH((k_sobel16<true, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_sobel16<false, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (rg == 2) {
        if (keep_cols) GS_LAUNCH((k_sobel16<true, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_sobel16<false, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (keep_cols) GS_LAUNCH(k_sobel16<true>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else GS_LAUNCH(k_sobel16<false>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
    } else {
      GS_LAUNCH(k_sobel_px, grid2d(w, h, nn), dim3(64, 4), 0, st, d, s, w, h, fb);
    }
  }
}

template <bool DILATE>
void launch_morph(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  if (n == 0) return;
  hipStream_t st = ctx().s();
  const size_t fb = (size_t)w * h;
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    uint8_t *d = dst + fb * f0;
    const uint8_t *s = src + fb * f0;
    if (strip_ok(w, h, d, s)) {
      const int rg = strip_mode(w, s);
      const StripCfg c = strip_cfg(w, h, nn, 5, 2, 4, rg);
      if (rg == 1) GS_LAUNCH((k_morph16<DILATE, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else if (rg == 2) GS_LAUNCH((k_morph16<DILATE, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else GS_LAUNCH(k_morph16<DILATE>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
    } else {
      GS_LAUNCH(k_morph_px<DILATE>, grid2d(w, h, nn), dim3(64, 4), 0, st, d, s, w, h, fb);
    }
  }
}

void launch_integral(const uint8_t *src, unsigned w, unsigned h, unsigned n, unsigned *ii) {
  if (n == 0) return;
  hipStream_t st = ctx().s();
  const size_t fp = (size_t)w * h;
  /* the banded form takes any width >= 32 at any alignment (round 4: ragged rows, frames wider than 4096 px in column
   * chunks); key 6 = 1 or key 21 = 1: the rows + columns form */
  const bool banded = g_tune[6] != 1 && fp * 4 < 0x7fffffffull &&
                      (g_tune[21] == 1 ? (w % 16 == 0 && w <= 4096 && al16(src) && al16(ii)) : w >= 32);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    if (banded) {
      const bool wide = w > 4096;
      unsigned nb = std::max(1u, 8u * topo().cus / nn);      /* ~8 blocks per CU in flight */
      nb = std::min(nb, std::max(1u, h / 8));
      if (wide) nb = std::max(nb, (h + kIntegralWideRows - 1) / kIntegralWideRows); /* a row's carry waits in LDS */
      const unsigned BH = (h + nb - 1) / nb;
      nb = (h + BH - 1) / BH;
      unsigned *cs = (unsigned *)ctx().scratch(SL_HISTP, (size_t)nn * nb * w * 4);
      const uint8_t *s = src + fp * f0;
      unsigned *o = ii + fp * f0;
      GS_LAUNCH(k_integral_colsum, dim3((w + 4095) / 4096, nb, nn), dim3(256), 0, st, s, w, h, BH, nb, cs);
      GS_LAUNCH(k_integral_colbase, dim3((w + 63) / 64, nn), dim3(64, 16), 0, st, cs, w, nb);
      const dim3 gw(1, (nb + 3) / 4, nn);
      const bool rg = (w & 3u) != 0u;
      if (g_tune[6] == 2 && !rg && !wide && w % 16 == 0) /* the block-per-band form (one barrier per row), kept for comparison */
        GS_LAUNCH(k_integral_band, dim3(1, nb, nn), dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (wide && rg) GS_LAUNCH((k_integral_wave<16, true, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (wide) GS_LAUNCH((k_integral_wave<16, false, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (w <= 2048 && rg) GS_LAUNCH((k_integral_wave<8, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (w <= 2048) GS_LAUNCH(k_integral_wave<8>, gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else if (rg) GS_LAUNCH((k_integral_wave<16, true>), gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
      else GS_LAUNCH(k_integral_wave<16>, gw, dim3(256), 0, st, s, w, h, BH, nb, (const unsigned *)cs, o);
    } else {
      GS_LAUNCH(k_integral_rows, dim3(h, nn), dim3(256), 0, st, src + fp * f0, w, h, ii + fp * f0);
      GS_LAUNCH(k_integral_cols, dim3((w + 255) / 256, nn), dim3(256), 0, st, ii + fp * f0, w, h);
    }
  }
}

/* gs_blur for any radius: register strips for r = 1..3 on aligned frames, otherwise clipped box
 * sums from a scratch integral image (exact: both are u32-modular like the reference). */
template <int MODE>
void launch_box_generic(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                        unsigned radius, int c) {
  hipStream_t st = ctx().s();
  const size_t fp = (size_t)w * h;
  const unsigned r = std::min(radius, std::max(w, h)); /* larger windows clip identically */
  if (g_tune[6] != 3 && r >= 1 && r <= 127 && w <= 4096 && strip_ok(w, h, dst, src)) {
    /* sliding box sums straight from the source rows (k_box.h): 3-4 B/px instead of the ~16 of the integral-image
     * route below (64 4K frames: 2.8 ms whatever the radius; this one: r = 16 0.36 ms, r = 40 0.63 ms); the kernel's
     * u16 column sums and LDS halo hold up to r = 127 */
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      /* r <= 16: the window's raw rows stay in registers (k_box16r: 2 B/px instead of 3-4; 64 x 4K: 0.22 ms for r <= 9,
       * 0.28 up to 16, against 0.31-0.42 -- profiles/r03r_box_ring.log).  Key 6 = 4: k_box16 always. */
      const bool ring = g_tune[6] != 4 && !ragged(w) && r <= box_ring_max() && w >= 32 && h >= 2 * r + 1 && (MODE == 0 || (c > -(1 << 30) && c < (1 << 30)));
      /* band height: the launch should be whole rounds of the blocks the chip holds (256 CUs x 4 of them, fewer for the
       * register-heavy ring kernels of r >= 8 / 10), and a band first loads 2r+1 rows it does not output -- loads and
       * adds only since the vertical-first form, ~0.3 of an output row each.  Pick the band count with the smallest
       * rounds x (T + 0.3 (2r+1)): 64 4K frames -> 16 bands of 135 rows (r = 16: 0.50 -> 0.36 ms, r = 40: 1.11 -> 0.63),
       * 8 frames -> 128 bands (r = 40: 1.0 -> 0.16 ms; profiles/r02l_box_T.log).  Round 1's "at least four window
       * heights per band" dates from a prologue that cost more than the rows it preceded. */
      const unsigned threads = w <= 1024 ? 64u : w <= 2048 ? 128u : 256u; /* a thread owns 16 px of the row */
      const unsigned slots = topo().cus * box_blocks_per_cu(MODE, ring ? r : 0u, threads);
      unsigned T = h;
      if (g_tune[0] > 0) {
        T = (unsigned)g_tune[0];
      } else {
        /* once the launch is several full rounds, more bands only add prologues (cost ~ nn h / slots + nn nbc k / slots
         * grows with nbc), so the search stops at 4 rounds' worth of blocks: at most 4096 candidates however tall
         * the image is; the last answer is kept per (h, nn, r, slots) */
        static thread_local struct { unsigned h, nn, r, slots, T; } memo = {0, 0, 0, 0, 0};
        if (memo.h == h && memo.nn == nn && memo.r == r && memo.slots == slots) {
          T = memo.T;
        } else {
          double best = 1e30;
          const unsigned nbc_max = std::max(1u, std::min(h / 8u, std::max(1u, 4u * slots / nn)));
          for (unsigned nbc = 1; nbc <= nbc_max; nbc++) {
            const unsigned t = (h + nbc - 1) / nbc;
            const double rounds = (double)(((unsigned long long)nn * ((h + t - 1) / t) + slots - 1) / slots);
            const double cost = rounds * ((double)t + (ring ? 0.12 : 0.3) * (2.0 * r + 1.0)); /* ring: 16 VALU per start-up row */
            if (cost < best - 1e-9) best = cost, T = t;
          }
          memo = {h, nn, r, slots, T};
        }
      }
      const unsigned nb = (h + T - 1) / T;
      launch_box(MODE, ring ? r : 0u, dim3(1, nb, nn), threads, st, dst + fp * f0, src + fp * f0, w, h, T, fp, r, c);
    }
    return;
  }
  const unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(n, (256u << 20) / (fp * 4) + 1));
  unsigned *ii = (unsigned *)ctx().scratch(SL_II, fp * 4 * group);
  for (unsigned f0 = 0; f0 < n; f0 += group) {
    const unsigned nn = std::min(group, n - f0);
    launch_integral(src + fp * f0, w, h, nn, ii);
    GS_LAUNCH(k_box_px<MODE>, grid2d(w, h, nn), dim3(64, 4), 0, st, dst + fp * f0, src + fp * f0,
              (const unsigned *)ii, w, h, r, c, fp);
  }
}

void launch_blur(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                 unsigned radius) {
  if (n == 0) return;
  hipStream_t st = ctx().s();
  const size_t fb = (size_t)w * h;
  if (radius >= 1 && radius <= 3 && strip_ok(w, h, dst, src) && h > 2 * radius && w > 2 * radius) {
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      uint8_t *d = dst + fb * f0;
      const uint8_t *s = src + fb * f0;
      const int rg = strip_mode(w, s);
      const StripCfg c = strip_cfg(w, h, nn, 5, radius, radius == 1 ? 4 : radius == 2 ? 6 : 12, rg);
      if (rg == 1) {
        if (radius == 1) GS_LAUNCH((k_blur16<1, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else if (radius == 2) GS_LAUNCH((k_blur16<2, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_blur16<3, 1>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (rg == 2) {
        if (radius == 1) GS_LAUNCH((k_blur16<1, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else if (radius == 2) GS_LAUNCH((k_blur16<2, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
        else GS_LAUNCH((k_blur16<3, 2>), c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      } else if (radius == 1) GS_LAUNCH(k_blur16<1>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else if (radius == 2) GS_LAUNCH(k_blur16<2>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      else GS_LAUNCH(k_blur16<3>, c.grid, c.block, 0, st, d, s, w, h, c.T, fb | c.xcd_flag);
      /* the 2*radius vertically clipped rows of each frame get their true divisors */
      GS_LAUNCH(k_blur_edge_rows, dim3((w + 255) / 256, 2 * radius, nn), dim3(256), 0, st, d, s, w, h,
                (int)radius, fb);
    }
    return;
  }
  if (radius == 0) { /* 1x1 window: identity (sum/1) */
    if (dst != src) GS_HIP(hipMemcpyAsync(dst, src, fb * n, hipMemcpyDeviceToDevice, st));
    return;
  }
  launch_box_generic<0>(dst, src, w, h, n, radius, 0);
}

/* ------------------------------------------------------------------ histogram / otsu / threshold */
/* blocks per frame of k_hist_partial.  A block pays for zeroing and folding its 32 KB of LDS counters and for
 * filling its load queue, so it should run ~64 trips of 16 B per lane (512 4K frames: 4.9 Tpx/s at 31 trips,
 * 5.8 at 63; profiles/r02i_hist.log, r02i_hist_trips.log); a handful of frames is spread over the CUs
 * (256 CUs x 5 resident blocks) down to 16 trips per block, at most 256 blocks per frame for k_hist_reduce. */
constexpr unsigned kHistThreads = 256;
unsigned hist_threads() { return g_tune[10] >= 1000 ? (unsigned)(g_tune[10] / 1000) * 256u : kHistThreads; } /* experiments: key 10 = 1000 * (threads / 256) + trips */
unsigned hist_bpf(size_t frame_bytes, unsigned n) {
  if (g_tune[11] > 0) return (unsigned)g_tune[11];
  const size_t bt = hist_threads();
  const size_t chunks = frame_bytes / 16 + 1, trips = g_tune[10] % 1000 > 0 ? (size_t)(g_tune[10] % 1000) : 64 * 256 / bt;
  const size_t by_size = (chunks + bt * trips - 1) / (bt * trips);
  const size_t by_fill = std::min<size_t>(std::min<size_t>((5u * topo().cus + n - 1) / n, chunks / (bt * 16)), 256); /* 5 resident blocks per CU */
  return (unsigned)std::max<size_t>(1, std::min<size_t>(std::max(by_size, by_fill), 2048));
}
void launch_hist_partial(dim3 grid, hipStream_t st, const uint8_t *img, size_t frame_bytes, unsigned *partial) {
  switch (hist_threads()) {
    case 512: GS_LAUNCH(k_hist_partial<512>, grid, dim3(512), 0, st, img, frame_bytes, partial); break;
    case 1024: GS_LAUNCH(k_hist_partial<1024>, grid, dim3(1024), 0, st, img, frame_bytes, partial); break;
    default: GS_LAUNCH(k_hist_partial<256>, grid, dim3(256), 0, st, img, frame_bytes, partial);
  }
}
void launch_histogram(const uint8_t *img, size_t frame_bytes, unsigned n, unsigned *hist) {
  if (n == 0) return;
  hipStream_t st = ctx().s();
  /* k_hist_partial addresses a frame with 32-bit offsets: count a huge image in pieces (key 12: piece size in
   * bytes, so that tests reach this path with small images) */
  const size_t piece_bytes = g_tune[12] > 0 ? (size_t)g_tune[12] : kHistMaxFrame;
  if (frame_bytes > piece_bytes) {
    const unsigned pieces = (unsigned)(frame_bytes / piece_bytes);
    const size_t rest = frame_bytes - (size_t)pieces * piece_bytes;
    const unsigned bpf = hist_bpf(piece_bytes, pieces), bpr = rest ? hist_bpf(rest, 1) : 0;
    unsigned *partial = (unsigned *)ctx().scratch(SL_HISTP, ((size_t)pieces * bpf + bpr) * 256 * 4);
    for (unsigned f = 0; f < n; f++) {
      const uint8_t *p = img + frame_bytes * f;
      launch_hist_partial(dim3(bpf, pieces), st, p, piece_bytes, partial);
      if (rest)
        launch_hist_partial(dim3(bpr, 1), st, p + (size_t)pieces * piece_bytes, rest, partial + (size_t)pieces * bpf * 256);
      GS_LAUNCH(k_hist_reduce, dim3(1), dim3(256), 0, st, (const unsigned *)partial, pieces * bpf + bpr,
                hist + (size_t)f * 256, 0u);
    }
    return;
  }
  const unsigned bpf = hist_bpf(frame_bytes, n);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    unsigned *partial = (unsigned *)ctx().scratch(SL_HISTP, (size_t)nn * bpf * 256 * 4);
    launch_hist_partial(dim3(bpf, nn), st, img + frame_bytes * f0, frame_bytes, partial);
    GS_LAUNCH(k_hist_reduce, dim3(nn), dim3(256), 0, st, (const unsigned *)partial, bpf,
              hist + (size_t)f0 * 256, 0u);
  }
}
void launch_threshold(uint8_t *img, size_t frame_bytes, unsigned n, const uint8_t *thr_dev,
                      unsigned thr_const, hipStream_t on = nullptr) {
  if (n == 0) return;
  hipStream_t st = on ? on : ctx().s();
  const size_t chunks = frame_bytes / 16 + 2;
  const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((chunks + 255) / 256, 2048));
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_threshold, dim3(bx, nn), dim3(256), 0, st, img + frame_bytes * f0, frame_bytes,
              thr_dev ? thr_dev + f0 : nullptr, thr_const);
  }
}
void launch_otsu(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist,
                 uint8_t *thr) {
  launch_histogram(img, (size_t)(w * h), n, hist);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_otsu, dim3(nn), dim3(256), 0, ctx().s(), hist + (size_t)f0 * 256, w * h, thr + f0,
              (const unsigned *)nullptr, 0u, 0u);
  }
}

/* ------------------------------------------------------------------ ordered compaction driver */
template <bool QUAD = false, class F>
void run_compaction(unsigned long long *mask, unsigned *cnt, unsigned nchunks, unsigned n,
                    unsigned cap, unsigned *totals_dev, F emit, hipStream_t on = nullptr, unsigned *pfx_in = nullptr) {
  hipStream_t st = on ? on : ctx().s();
  if (nchunks <= kEmitSelfScan) { /* FAST on video frames, match: one launch less on a latency-bound tail */
    GS_LAUNCH((k_emit<F, QUAD>), dim3((nchunks + 3) / 4, n), dim3(256), 0, st,
              (const unsigned long long *)mask, (const unsigned *)cnt, (const unsigned *)nullptr, nchunks,
              cap, emit, totals_dev);
    return;
  }
  unsigned *pfx = pfx_in ? pfx_in : (unsigned *)ctx().scratch(SL_PFX, (size_t)n * nchunks * 4);
  GS_LAUNCH(k_chunk_scan, dim3(n), dim3(1024), 0, st, (const unsigned *)cnt, nchunks, pfx,
            totals_dev, cap);
  GS_LAUNCH((k_emit<F, QUAD>), dim3((nchunks + 3) / 4, n), dim3(256), 0, st,
            (const unsigned long long *)mask, (const unsigned *)cnt, (const unsigned *)pfx, nchunks,
            cap, emit, (unsigned *)nullptr);
}

/* ------------------------------------------------------------------ FAST */
/* gs_fast pass 1 (w, h >= 7, n <= kMaxZ): the LDS-tile kernel by default.  The strip kernel (gsh_tune key 7 = 1)
 * decides per 256-px row span instead of per 64 px: on 32 x 720p (profiles/r02i_fast_tile.log) it is 1.25x faster on
 * flat frames (1.6 vs 2.0 us per frame), equal on bright frames and 1.1-1.4x SLOWER on texture and on frames with
 * large p < t regions (the block-noise frames of configs[3]: there every pixel is a candidate under the reference's
 * unsigned wrap, and a 256-px span almost always touches one).  Key 7 = 2: one global byte load per ring pixel
 * (the round-1 form: texture-addresser bound, 4.9 vs 4.2 us per frame). */
/* zero_words / zero_n: words the default kernel clears on the side (pass 2's chunk counters); returns whether it did */
bool launch_fast_score(hipStream_t on, const uint8_t *img, uint8_t *score, unsigned w, unsigned h, unsigned n,
                       unsigned threshold, unsigned *zero_words = nullptr, unsigned zero_n = 0) {
  const size_t fb = (size_t)w * h;
  if (g_tune[7] == 1 && w % 4 == 0 && fb < 0x7fffffffull && ((uintptr_t)img & 3) == 0 && ((uintptr_t)score & 3) == 0 &&
      threshold <= 0xffffff00u) {
    /* strip kernel: ~6 waves per SIMD when the batch allows, bands of >= 8 rows */
    const unsigned cw = (w + 255) / 256, rows = h - 6;
    unsigned long long T = ((unsigned long long)rows * cw * n + 6143) / 6144;
    T = T < 8 ? 8 : T > 64 ? 64 : T;
    const unsigned nb = (rows + (unsigned)T - 1) / (unsigned)T;
    GS_LAUNCH(k_fast_score4, dim3(cw, (nb + 3) / 4, n), dim3(64, 4), 0, on, img, score, w, h, (unsigned)T, fb, threshold);
  } else if (g_tune[7] == 2) {
    GS_LAUNCH(k_fast_score_px, grid2d(w - 6, h - 6, n), dim3(64, 4), 0, on, img, score, w, h, fb, threshold);
  } else if (g_tune[7] == 3) {
    /* LDS tile + block-local candidate queue: measured and NOT the default (profiles/r03i_fast_candidate_queue.log, 32 x 720p
     * score pass): block noise 72.7 -> 69.9 us, tiled lena 109 -> 92, but +8 % on bright noise, flat and random frames --
     * half of the pass is the tile load, the compass filter and the byte stores, which the queue does not touch */
    GS_LAUNCH(k_fast_score_cq, dim3((w - 6 + 63) / 64, (h - 6 + kFastTileRows - 1) / kFastTileRows, n), dim3(64, 4), 0, on,
              img, score, w, h, fb, threshold);
  } else if (g_tune[7] == 4 || threshold > 0xffffff00u) { /* round-2 default: one pixel per lane, whole wave rows scored */
    GS_LAUNCH(k_fast_score_tile, dim3((w - 6 + 63) / 64, (h - 6 + kFastTileRows - 1) / kFastTileRows, n), dim3(64, 4), 0, on,
              img, score, w, h, fb, threshold);
  } else { /* LDS tile, 4 px per thread through the compass filter, candidates queued (k_fast.h) */
    const unsigned tx = (w - 6 + 63) / 64, ty = (h - 6 + kFastTileRows - 1) / kFastTileRows;
    const unsigned long long nt = (unsigned long long)tx * ty * n;
    GS_ASSERT(nt <= 0x7ffffff0ull); /* 2^31 tiles = 2^41 pixels in one call */
    const unsigned share = (g_tune[18] == 1 || !topo().eight_xcds()) ? 0u : (unsigned)((nt + 7) / 8); /* key 18 = 1: tiles in launch order */
    GS_LAUNCH(k_fast_score_q4, dim3(share ? share * 8u : (unsigned)nt), dim3(64, 4), 0, on, img, score, w, h, fb, threshold, tx,
              ty, (unsigned)nt, share, zero_words, zero_n);
    return true;
  }
  return false;
}

/* clip_w / clip_h (single frame only): the caller's score map is smaller than the image; positions
 * outside it read 0 in the NMS pass like gs_get does (ref :524) */
void launch_fast(const uint8_t *img, uint8_t *score, unsigned w, unsigned h, unsigned n,
                 unsigned *kps, unsigned *counts, unsigned nkps, unsigned threshold, unsigned clip_w = 0,
                 unsigned clip_h = 0) {
  hipStream_t st = ctx().s();
  if (n == 0) return;
  if (n > kMaxZ) { /* grid.y / grid.z carry the frame index: split like every other launcher */
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ)
      launch_fast(img + (size_t)w * h * f0, score + (size_t)w * h * f0, w, h, std::min(kMaxZ, n - f0),
                  kps + (size_t)f0 * nkps * 12, counts + f0, nkps, threshold);
    return;
  }
  if (w < 7 || h < 7) { /* reference loops are empty for 3 <= dim < 7 */
    GS_HIP(hipMemsetAsync(counts, 0, (size_t)n * 4, st));
    return;
  }
  const size_t fb = (size_t)w * h;
  /* key 7 = 6: both passes in one walk (k_fast_fused.h).  Measured and NOT the default: 32 x 720p block noise 127 us against
   * 63 + 23 for the two passes below -- the walk executes 50 M VALU + 21 M SALU wave-instructions where the two passes
   * execute 40 M + 18 M (ownership / interior masks, the border handling of the score tile, three barriers per step) at
   * 4 waves per SIMD instead of 8 (profiles/r03t_fast_fused_not_kept.log, r03u_pmc_fast_fused.txt). */
  if (g_tune[7] == 6 && g_tune[19] != 1 && threshold <= 0xffffff00u && !(clip_w && n == 1 && (clip_w < w || clip_h < h)) &&
      (unsigned long long)((w + 63) / 64) * 64 * h < (1ull << 32)) {
    const unsigned wpr = (w + 63) / 64, nwords = wpr * h, nchunks = (nwords + kChunkWords - 1) / kChunkWords;
    unsigned long long *mask = (unsigned long long *)ctx().scratch(SL_MASK, (size_t)n * nchunks * kChunkWords * 8);
    unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, (size_t)n * nchunks * 4);
    unsigned *pfx = (unsigned *)ctx().scratch(SL_PFX, (size_t)n * nchunks * 4);
    GS_HIP(hipMemsetAsync(mask, 0, (size_t)n * nchunks * kChunkWords * 8, st));
    GS_HIP(hipMemsetAsync(cnt, 0, (size_t)n * nchunks * 4, st));
    /* bands of 16 m - 2 rows: ~4 rounds of the 2048 blocks the chip holds, at least 30 rows (2 of every 16 m score rows
     * are computed twice), at most 254 */
    const unsigned strips = (w - 6 + kFfCols - 1) / kFfCols, rows = h - 6;
    const unsigned long long cols = (unsigned long long)strips * n;
    const unsigned want = (unsigned)std::max<unsigned long long>(1, 8192 / cols);
    unsigned m = g_tune[0] > 0 ? (unsigned)g_tune[0] : (rows / want + 2 + 15) / 16;
    m = std::max(2u, std::min(16u, m));
    const unsigned bands = (rows + 16 * m - 3) / (16 * m - 2);
    const unsigned long long nt = cols * bands;
    GS_ASSERT(nt <= 0x7ffffff0ull);
    const unsigned share = (g_tune[18] == 1 || !topo().eight_xcds()) ? 0u : (unsigned)((nt + 7) / 8);
    FastFusedArgs fa{img, score, w, h, fb, threshold, strips, bands, m, (unsigned)nt, share, mask, cnt, wpr, nchunks};
    GS_LAUNCH(k_fast_fused, dim3(share ? share * 8u : (unsigned)nt), dim3(64, 4), 0, st, fa);
    run_compaction(mask, cnt, nchunks, n, nkps, counts,
                   FastEmitPadded{score, w, wpr * 64u, fb, kps, nkps, ((uintptr_t)kps & 15) == 0}, st, pfx);
    return;
  }
  /* pass 2 in strip form (k_fast_nms.h) when the score map qualifies for the strip machinery: items numbered over
   * rows padded to whole mask words.  Key 19 = 1: the item-by-item kernel k_fast_nms (round 2). */
  if (g_tune[19] != 1 && strip_ok(w, h, score, score) && w >= 32 && (unsigned long long)((w + 63) / 64) * 64 * h < (1ull << 32)) {
    const unsigned wpr = (w + 63) / 64, nwords = wpr * h, nchunks = (nwords + kChunkWords - 1) / kChunkWords;
    unsigned long long *mask = (unsigned long long *)ctx().scratch(SL_MASK, (size_t)n * nchunks * kChunkWords * 8);
    unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, (size_t)n * nchunks * 4);
    unsigned *pfx = (unsigned *)ctx().scratch(SL_PFX, (size_t)n * nchunks * 4);
    /* the default score kernel zeroes the chunk counters on the side (a fill launch costs 5 us of a 100-us call) */
    GS_ASSERT((unsigned long long)n * nchunks < (1ull << 32)); /* n <= 65535 frames of < 2^32 padded items */
    const bool zeroed = launch_fast_score(st, img, score, w, h, n, threshold, cnt, n * nchunks);
    if (clip_w && n == 1 && (clip_w < w || clip_h < h))
      GS_LAUNCH(k_fast_clip, grid2d(w, h, 1), dim3(64, 4), 0, st, score, w, h, clip_w, clip_h);
    if (!zeroed) GS_HIP(hipMemsetAsync(cnt, 0, (size_t)n * nchunks * 4, st));
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      const StripCfg c = strip_cfg(w, h - 6, nn);
      GS_LAUNCH(k_fast_nms16, c.grid, c.block, 0, st, (const uint8_t *)score + fb * f0, w, h, c.T, fb | c.xcd_flag,
                mask + (size_t)f0 * nchunks * kChunkWords, cnt + (size_t)f0 * nchunks, wpr, nchunks);
    }
    /* (Tried and not kept: eight chunks per emit wave -- fewer waves, loads batched -- 19 -> 33 us per 32 x 720p: the pass is
     * one wave's latency chain, and 14,464 small waves hide it better than 1,808 long ones.  Bands of 16 rows for the NMS
     * kernel: 23.6 -> 25 us.  profiles/r03m_fast_emit_not_kept.log) */
    run_compaction(mask, cnt, nchunks, n, nkps, counts,
                   FastEmitPadded{score, w, wpr * 64u, fb, kps, nkps, ((uintptr_t)kps & 15) == 0}, st, pfx);
    return;
  }
  launch_fast_score(st, img, score, w, h, n, threshold);
  const unsigned nitems = (w - 6) * (h - 6);
  const unsigned nchunks = (nitems + kChunkItems - 1) / kChunkItems;
  unsigned long long *mask =
      (unsigned long long *)ctx().scratch(SL_MASK, (size_t)n * nchunks * kChunkWords * 8);
  unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, (size_t)n * nchunks * 4);
  unsigned *pfx = (unsigned *)ctx().scratch(SL_PFX, (size_t)n * nchunks * 4);
  /* row = item / (w-6) by multiplication where the magic fits (see div_by) */
  const unsigned iw = w - 6;
  const unsigned magic = (iw > 256 && iw <= 8192 && (unsigned long long)iw * (h - 6) <= (1ull << 26))
                             ? (unsigned)(((1ull << 40) + iw - 1) / iw) : 0u;
  /* NMS flags, chunk scan, ordered emit of frames [f0, f0 + nn) on stream `on` */
  auto rest = [&](hipStream_t on, unsigned f0, unsigned nn) {
    if (clip_w && n == 1 && (clip_w < w || clip_h < h))
      GS_LAUNCH(k_fast_clip, grid2d(w, h, 1), dim3(64, 4), 0, on, score, w, h, clip_w, clip_h);
    unsigned *c = cnt + (size_t)f0 * nchunks; /* k_fast_nms stores every chunk's count: no zeroing */
    GS_LAUNCH(k_fast_nms, dim3(nchunks, nn), dim3(256), 0, on, (const uint8_t *)score + fb * f0, w, h, fb,
              mask + (size_t)f0 * nchunks * kChunkWords, c, nchunks, magic);
    run_compaction</*QUAD=*/true>(mask + (size_t)f0 * nchunks * kChunkWords, c, nchunks, nn, nkps, counts + f0, /* k_fast_nms: 4 items per lane */
                                  FastEmit{score + fb * f0, w, fb, kps + (size_t)f0 * nkps * 12, nkps, ((uintptr_t)kps & 15) == 0}, on,
                                  pfx + (size_t)f0 * nchunks);
  };
  /* Tried and not kept: cutting a batch into 2-8 groups of frames and running group i's NMS / scan / emit on the side
   * stream under group i+1's score pass.  The cross-stream event hops cost more than the ~60 us of small passes
   * they could hide: 32 x 720p 4.2 us per frame in one piece, 5.1 in two, 6.8 in four (profiles/r02i_fast_groups_not_kept.log). */
  rest(st, 0, n);
}

/* ------------------------------------------------------------------ LBP cascade */
}  // namespace

struct gsh_cascade {
  unsigned window_w, window_h, nfeatures, nweaks, nstages, nsub = 0;
  std::vector<int8_t> features;
  std::vector<uint16_t> weak_feature_idx;
  LbpWeak *d_weak = nullptr;
  LbpStage *d_stage = nullptr;
  int32_t *d_subsets = nullptr;
  unsigned *d_pass_lut = nullptr; /* per stage: truth table of the stage decision over its match bits (k_lbp_dense.h) */
  unsigned pre_max = 0;           /* leading stages with <= kPreMaxWeaks weak classifiers */
  unsigned long long id = 0; /* unique per handle: key of the calling threads' geometry caches */
};

namespace {
/* the device tables of a handle, without touching any context (used while a context is being released) */
void gsh_cascade_tables_deleter::operator()(gsh_cascade *dc) const {
  (void)hipFree(dc->d_weak), (void)hipFree(dc->d_stage), (void)hipFree(dc->d_subsets), (void)hipFree(dc->d_pass_lut);
  delete dc;
}
}  // namespace

namespace {

/* The reference's scale loop and per-feature truncation (ref :819-821, :799-804), float32. */
void build_scales(const gsh_cascade &c, unsigned iw, unsigned ih, float scale_factor,
                  float min_scale, float max_scale, int step, std::vector<LbpScale> &scales,
                  std::vector<LbpGeom> &geom, bool &guard, unsigned long long &nwin,
                  unsigned long long *max_cell_px = nullptr) {
  scales.clear();
  geom.clear();
  guard = false;
  nwin = 0;
  unsigned chunk_base = 0;
  const unsigned S = iw + 1;
  for (float scale = min_scale; scale <= max_scale; scale *= scale_factor) {
    const int win_w = (int)((int)c.window_w * scale), win_h = (int)((int)c.window_h * scale);
    if (win_w > (int)iw || win_h > (int)ih) break;
    LbpScale sc;
    sc.win_w = win_w, sc.win_h = win_h;
    sc.nx = ((unsigned)((int)iw - win_w)) / (unsigned)step + 1;
    sc.ny = ((unsigned)((int)ih - win_h)) / (unsigned)step + 1;
    sc.chunk_base = chunk_base;
    sc.nchunks = (sc.nx * sc.ny + kChunkItems - 1) / kChunkItems;
    chunk_base += sc.nchunks;
    nwin += (unsigned long long)sc.nx * sc.ny;
    for (unsigned wi = 0; wi < c.nweaks; wi++) {
      const int fi = c.weak_feature_idx[wi];
      int fx = (int)((int)c.features[fi * 4 + 0] * scale);
      int fy = (int)((int)c.features[fi * 4 + 1] * scale);
      int fw = (int)((int)c.features[fi * 4 + 2] * scale);
      int fh = (int)((int)c.features[fi * 4 + 3] * scale);
      if (fw < 1) fw = 1;
      if (fh < 1) fh = 1;
      if (fx < 0 || fy < 0 || fx + 3 * fw > win_w || fy + 3 * fh > win_h) guard = true;
      geom.push_back(LbpGeom{(fy * (int)S + fx) * 4, fw * 4, fh * (int)S * 4, fh});
      if (max_cell_px) *max_cell_px = std::max(*max_cell_px, (unsigned long long)fw * (unsigned long long)fh);
    }
    scales.push_back(sc);
    if (scales.size() >= 4096 || !(scale_factor > 1.0f)) break; /* the reference would not terminate */
  }
}

LbpGeomCache &cascade_prepare(const gsh_cascade *dc, unsigned iw, unsigned ih, float sf, float mn, float mx,
                              int step) {
  Ctx &cx = ctx();
  if (cx.geom_cache.size() > 16 && !cx.geom_cache.count(dc->id)) { /* handles come and go: bound the cache */
    cx.sync();
    cx.drop_geom();
  }
  LbpGeomCache &gc = cx.geom_cache[dc->id];
  if (gc.iw == iw && gc.ih == ih && gc.sf == sf && gc.mn == mn && gc.mx == mx && gc.step == step && gc.d_scales)
    return gc;
  std::vector<LbpGeom> geom;
  gc.max_cell_px = 0;
  build_scales(*dc, iw, ih, sf, mn, mx, step, gc.scales, geom, gc.guard, gc.nwindows, &gc.max_cell_px);
  cx.sync(); /* tables may be in use by an earlier launch of this thread */
  const size_t sb = std::max<size_t>(1, gc.scales.size()) * sizeof(LbpScale);
  const size_t gb = std::max<size_t>(1, geom.size()) * sizeof(LbpGeom);
  if (gc.d_scales_cap < sb) {
    if (gc.d_scales) GS_HIP(hipFree(gc.d_scales));
    GS_HIP(hipMalloc((void **)&gc.d_scales, sb));
    gc.d_scales_cap = sb;
  }
  if (gc.d_geom_cap < gb) {
    if (gc.d_geom) GS_HIP(hipFree(gc.d_geom));
    GS_HIP(hipMalloc((void **)&gc.d_geom, gb));
    gc.d_geom_cap = gb;
  }
  if (!gc.scales.empty()) {
    GS_HIP(hipMemcpy(gc.d_scales, gc.scales.data(), gc.scales.size() * sizeof(LbpScale), hipMemcpyHostToDevice));
    GS_HIP(hipMemcpy(gc.d_geom, geom.data(), geom.size() * sizeof(LbpGeom), hipMemcpyHostToDevice));
  }
  gc.total_chunks = 0, gc.max_chunks = 0;
  for (auto &sc : gc.scales) {
    gc.total_chunks += sc.nchunks;
    gc.max_chunks = std::max(gc.max_chunks, sc.nchunks);
  }
  /* prefilter layout: one bit per window, window rows padded to whole u64 words; 64 x 64-window tiles */
  gc.pre_words = 0, gc.max_tiles = 0;
  if (step == 1 && !gc.scales.empty()) {
    std::vector<LbpPreScale> pre;
    for (auto &sc : gc.scales) {
      LbpPreScale ps;
      ps.word_base = gc.pre_words, ps.wpr = (sc.nx + 63u) / 64u, ps.tiles_x = ps.wpr;
      ps.ntiles = ps.tiles_x * ((sc.ny + kPreTile - 1u) / kPreTile), ps.pad = 0;
      gc.pre_words += (unsigned long long)ps.wpr * sc.ny;
      gc.max_tiles = std::max(gc.max_tiles, ps.ntiles);
      pre.push_back(ps);
    }
    const size_t pb = pre.size() * sizeof(LbpPreScale);
    if (gc.d_pre_cap < pb) {
      if (gc.d_pre) GS_HIP(hipFree(gc.d_pre));
      GS_HIP(hipMalloc((void **)&gc.d_pre, pb));
      gc.d_pre_cap = pb;
    }
    GS_HIP(hipMemcpy(gc.d_pre, pre.data(), pb, hipMemcpyHostToDevice));
  }
  gc.iw = iw, gc.ih = ih, gc.sf = sf, gc.mn = mn, gc.mx = mx, gc.step = step;
  return gc;
}

/* padded: n frames of (iw+1)*(ih+1) u32 on device */
void launch_lbp_padded(const gsh_cascade *dc, const LbpGeomCache &gc, const unsigned *padded, unsigned iw,
                       unsigned ih, unsigned n, unsigned *rects, unsigned *counts, unsigned max_rects,
                       int step, const unsigned *not_integral) {
  hipStream_t st = ctx().s();
  if (gc.scales.empty() || max_rects == 0) {
    GS_HIP(hipMemsetAsync(counts, 0, (size_t)n * 4, st));
    return;
  }
  const unsigned nch = gc.total_chunks;
  unsigned long long *mask =
      (unsigned long long *)ctx().scratch(SL_MASK, (size_t)n * nch * kChunkWords * 8);
  const unsigned nsc0 = (unsigned)gc.scales.size();
  /* chunk counters, then the early-exit counters: one per group of 32 chunks, one per 1024 */
  const unsigned ngroups = (nch >> kLbpGroupShift) + 1, nsupers = (nch >> kLbpSuperShift) + 1;
  const size_t ncnt = (size_t)n * nch + (size_t)n * ngroups + (size_t)n * nsupers + n;
  unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, ncnt * 4);
  GS_HIP(hipMemsetAsync(cnt, 0, ncnt * 4, st));
  GS_HIP(hipMemsetAsync(mask, 0, (size_t)n * nch * kChunkWords * 8, st));
  LbpArgs a;
  a.hits_group = cnt + (size_t)n * nch;
  a.hits_super = a.hits_group + (size_t)n * ngroups;
  a.hits_total = a.hits_super + (size_t)n * nsupers;
  a.scale0 = 0;
  a.ngroups = ngroups, a.nsupers = nsupers;
  a.evaluated = g_lbp_evaluated;
  a.nscales = nsc0, a.cap = max_rects;
  a.nwindows_cap = (unsigned)std::min<unsigned long long>(gc.nwindows, 0xffffffffull);
  a.padded = padded;
  a.frame_stride = (size_t)(iw + 1) * (ih + 1);
  a.S = iw + 1;
  a.limit_bytes = (unsigned)((a.frame_stride - 1) * 4);
  a.step = step;
  a.nweaks = dc->nweaks, a.nstages = dc->nstages, a.nsub = dc->nsub;
  a.scales = gc.d_scales, a.geom = gc.d_geom, a.weak = dc->d_weak, a.stage = dc->d_stage;
  a.subsets = dc->d_subsets;
  a.mask = mask, a.chunk_count = cnt, a.total_chunks = nch;
  a.pre_bitmap = nullptr, a.pre_scales = gc.d_pre, a.pre_words = gc.pre_words, a.pre_stages = 0;
  const unsigned nsc = (unsigned)gc.scales.size();
  /* XCD-aware chunk mapping (k_lbp.h) once the integral image no longer fits one XCD's 4 MB L2: 1080p -3 %, 4K block
   * noise -4 %, 4K edge maps -12 % (5.76 -> 5.08 ms per frame); 720p (3.7 MB) is 1-4 % better off in dispatch order
   * (profiles/r02l_lbp_xcd.log).  Key 13: 1 = never, 2 = always. */
  a.xcd_swizzle = g_tune[13] == 1 ? 0u : g_tune[13] == 2 ? 1u : ((a.frame_stride * 4 >= (size_t)6 << 20 && topo().eight_xcds()) ? 1u : 0u);
  const size_t lds = (size_t)dc->nstages * sizeof(LbpStage) +
                     (size_t)dc->nweaks * (sizeof(LbpWeak) + sizeof(LbpGeom)) + (size_t)dc->nsub * 4;
  GS_ASSERT(lds <= 60 * 1024 && "cascade tables must fit the block's LDS");
  /* phases of the block-local survivor re-packing: g_tune[4] selects a preset */
  LbpPhases ph;
  {
    static const unsigned presets[8][kLbpMaxPhases] = {
        {2, 4, 7, 99, 99, 99, 99, 99},    /* 0 default (measured best on MI355X, frontalface) */
        {99, 99, 99, 99, 99, 99, 99, 99}, /* 1: single dense phase (no re-packing) */
        {1, 2, 4, 6, 9, 13, 99, 99},
        {1, 2, 3, 4, 6, 8, 12, 99},
        {1, 2, 5, 99, 99, 99, 99, 99},
        {2, 5, 99, 99, 99, 99, 99, 99},
        {2, 6, 99, 99, 99, 99, 99, 99},
        {3, 7, 99, 99, 99, 99, 99, 99}};
    unsigned custom[kLbpMaxPhases];
    const unsigned *pr = presets[(g_tune[4] >= 0 && g_tune[4] < 8) ? g_tune[4] : 0];
    if (g_tune[4] >= 1000) { /* experiments: 1000 + e0 + 32 e1 + 1024 e2 + 32768 e3 (0 = no further split) */
      unsigned v = (unsigned)g_tune[4] - 1000u;
      for (unsigned i = 0; i < kLbpMaxPhases; i++, v >>= 5) custom[i] = (v & 31u) ? (v & 31u) : 99u;
      pr = custom;
    }
    ph.n = 0;
    unsigned prev = 0;
    for (unsigned i = 0; i < kLbpMaxPhases && prev < dc->nstages; i++) {
      const unsigned e = (i + 1 == kLbpMaxPhases) ? dc->nstages : std::min(pr[i], dc->nstages);
      if (e <= prev) continue;
      ph.end[ph.n++] = e, prev = e;
    }
    if (ph.n == 0) ph.n = 1, ph.end[0] = dc->nstages;
    ph.end[ph.n - 1] = dc->nstages;
    /* preset 0 (default): first re-packing point chosen per block between stages 2 and 8 (k_lbp.h) */
    ph.adaptive_max = (g_tune[4] == 0 && dc->nstages > 2) ? 8u : 0u; /* 6 .. 15 within 1.5 % (profiles/r02l_lbp_adaptive_xcd.log) */
    ph.adaptive_tenths = 2u;
    ph.quad = g_tune[17] == 1 ? 0u : 1u; /* key 17 = 1: one lane per re-packed window (the round-2 form) */
    /* with quad-lane survivors (profiles/r03f_lbp_adaptive_quad.log, r03k_lbp_adaptive_next.log): +1 +2 +4 is best on
     * block noise (8 x 1080p 0.76 vs 0.80 ms for +1 +3 +6, 4K 3.15 vs 3.17) and within 0.5 % of the best on edge maps */
    ph.adaptive_next[0] = 1u, ph.adaptive_next[1] = 2u, ph.adaptive_next[2] = 4u;
    if (ph.adaptive_max && g_tune[9] > 0) { /* experiments: key 9 = max + 16 * tenths (+ 256 d1 + 4096 d2 + 65536 d3: later points) */
      const unsigned v = (unsigned)g_tune[9];
      ph.adaptive_max = v & 15u, ph.adaptive_tenths = (v >> 4) & 15u;
      if (v >> 8) ph.adaptive_next[0] = (v >> 8) & 15u, ph.adaptive_next[1] = (v >> 12) & 15u, ph.adaptive_next[2] = (v >> 16) & 15u;
    }
  }
  const size_t lds_all = ((lds + 15) & ~(size_t)15) + 2 * kChunkItems * 2 + 64 * 4 + 16;
  /* Prefilter (k_lbp_dense.h, OPTIONAL, key 14 = k > 0: that many stages; default off): stages [0, pre) for every
   * window with the table rows shared down the columns of 64 x 64-window tiles; the cascade kernel then starts from
   * the surviving set.  Needs unit step, in-window geometry, the adaptive preset and stages of <= 5 weak classifiers.
   * Measured (profiles/r03a_lbp_prefilter_first.log, r03d_pmc_lbp.txt): 2.8x fewer gather instructions per weak
   * classifier, but 0.22 ms per classifier and 4K frame against 0.18 for the dense phase of k_lbp_cascade -- ~1000
   * tiles per XCD walk 64-row bands of the table at once, half their L2 requests miss (the cascade kernel's chunks
   * stay inside a 3 MB band: 1 % misses) and the texture path stalls on pending misses half the time.
   * The prefilter cannot see detections of its own launch, so the scales are issued in GROUPS (prefilter, then
   * cascade, group after group on the stream): a group's tiles skip once the groups before it hold max_rects
   * detections -- the reference stops scanning there (ref :819-823) -- while the chunk-granular exit inside the
   * cascade kernel stays as it was.  A group is at least ~16 M windows (key 15 overrides, a test hook), so a
   * launch always fills the chip: 8 x 4K = one scale per group, one 1080p frame = two groups. */
  const int k14 = g_tune[14] >= 100 ? g_tune[14] - 100 : g_tune[14];
  unsigned pre = k14 > 0 ? (unsigned)k14 : 0u; /* off by default: measured slower than the dense phase it replaces (see above) */
  pre = std::min(pre, std::min(dc->pre_max, dc->nstages > 0 ? dc->nstages - 1u : 0u));
  if (step != 1 || gc.guard || !ph.adaptive_max || !gc.d_pre || !dc->d_pass_lut || a.frame_stride * 4 >= (1ull << 31)) pre = 0;
  LbpPreArgs pa;
  pa.pass_lut = dc->d_pass_lut, pa.bitmap = nullptr, pa.xcd_swizzle = a.xcd_swizzle;
  pa.not_integral = not_integral;
  /* sign-bit compares (k_lbp_dense.h: lbp_code8) need every cell sum < 2^31: cells of the prefiltered classifiers
   * cover at most max_cell_px pixels of <= 255 each.  Key 14 + 100 forces the general compare (A/B, tests). */
  pa.small_cells = (not_integral && gc.max_cell_px < (1ull << 31) / 255ull && g_tune[14] < 100) ? 1u : 0u;
  if (pre) {
    pa.bitmap = (unsigned long long *)ctx().scratch(SL_PRE, (size_t)n * gc.pre_words * 8);
    a.pre_bitmap = pa.bitmap, a.pre_stages = pre;
  }
  const unsigned long long group_windows = g_tune[15] > 0 ? (unsigned long long)g_tune[15] : 16ull << 20;
  for (unsigned s0 = 0; s0 < nsc;) {
    unsigned s1 = s0;
    unsigned long long wsum = 0;
    unsigned mc = 0, mt = 0;
    do {
      wsum += (unsigned long long)n * gc.scales[s1].nx * gc.scales[s1].ny;
      mc = std::max(mc, gc.scales[s1].nchunks);
      mt = std::max(mt, (gc.scales[s1].nx + 63u) / 64u * ((gc.scales[s1].ny + kPreTile - 1u) / kPreTile));
      s1++;
    } while (pre && s1 < nsc && wsum < group_windows);
    if (!pre) /* no prefilter: one launch over all scales, as before */
      for (; s1 < nsc; s1++) mc = std::max(mc, gc.scales[s1].nchunks);
    a.scale0 = s0;
    if (pre) {
      const unsigned nblk = (mt + 3u) / 4u;
      const dim3 gd(pa.xcd_swizzle ? (nblk + 7u) & ~7u : nblk, s1 - s0, n);
      if (a.evaluated) GS_LAUNCH(k_lbp_dense<true>, gd, dim3(256), lds + 16, st, a, pa);
      else GS_LAUNCH(k_lbp_dense<false>, gd, dim3(256), lds + 16, st, a, pa);
    }
    const dim3 g(a.xcd_swizzle ? (mc + 7u) & ~7u : mc, s1 - s0, n);
    if (a.evaluated) { /* counting build: the same kernel + one register that counts classifier evaluations */
      if (gc.guard) GS_LAUNCH((k_lbp_cascade<true, true>), g, dim3(256), lds_all, st, a, ph);
      else GS_LAUNCH((k_lbp_cascade<false, true>), g, dim3(256), lds_all, st, a, ph);
    } else if (gc.guard) GS_LAUNCH(k_lbp_cascade<true>, g, dim3(256), lds_all, st, a, ph);
    else GS_LAUNCH(k_lbp_cascade<false>, g, dim3(256), lds_all, st, a, ph);
    s0 = s1;
  }
  if (g_tune[16] == 1) return; /* timing aid (scripts/bench_lbp_stages.py): the cascade kernels alone, no rect emission */
  run_compaction(mask, cnt, nch, n, max_rects, counts,
                 LbpEmit{gc.d_scales, (unsigned)gc.scales.size(), step, rects, max_rects});
}

constexpr unsigned kLbpGroup = 8; /* frames per cascade launch (bounds mask/padded scratch) */

void launch_lbp_unpadded(const gsh_cascade *dc, const unsigned *ii, unsigned iw, unsigned ih, unsigned n,
                         unsigned *rects, unsigned *counts, unsigned max_rects, float sf, float mn,
                         float mx, int step) {
  GS_ASSERT(step > 0);
  const LbpGeomCache &gc = cascade_prepare(dc, iw, ih, sf, mn, mx, step);
  hipStream_t st = ctx().s();
  const size_t fp = (size_t)iw * ih, pp = (size_t)(iw + 1) * (ih + 1);
  for (unsigned f0 = 0; f0 < n; f0 += kLbpGroup) {
    const unsigned nn = std::min(kLbpGroup, n - f0);
    unsigned *padded = (unsigned *)ctx().scratch(SL_PAD, pp * 4 * nn);
    unsigned *notii = (unsigned *)ctx().scratch(SL_NOTII, (size_t)nn * 4);
    GS_HIP(hipMemsetAsync(notii, 0, (size_t)nn * 4, st));
    GS_LAUNCH(k_integral_pad, dim3((iw + 64) / 64, (ih + 4) / 4, nn), dim3(64, 4), 0, st,
              ii + fp * f0, iw, ih, padded, notii);
    launch_lbp_padded(dc, gc, padded, iw, ih, nn, rects + (size_t)f0 * max_rects * 4, counts + f0,
                      max_rects, step, notii);
  }
}

/* one window on a (win_w+1) x (win_h+1) table: grid 1, block 64, lane 0 decides */
__global__ void k_lbp_single(LbpArgs a, const LbpGeom *geom, unsigned *out) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  GS_DYN_LDS(smem);
  const LbpLds t = lbp_stage_tables(smem, a, geom, threadIdx.x, 64u);
  __syncthreads();
  if (threadIdx.x == 0)
    out[0] = lbp_window_stages<true>(t, a.padded, 0u, a.limit_bytes, 0u, a.nstages) ? 1u : 0u;
}

/* ------------------------------------------------------------------ ORB host logic */

/* gs_orb_extract (ref :651-669) for up to 4 independent images (pyramid levels) with TWO host round
 * trips in total: (1) FAST + NMS + ordered emit + disc moments of every candidate slot, per level,
 * then one copy-back of counts + records + moments; host: stable sort (desc response, ref :639),
 * 15-px border filter, atan2f / sinf from libm (ref :100-101); (2) BRIEF for the kept keypoints of
 * all levels, one copy-back of the descriptors. */
struct OrbLevel {
  const uint8_t *img;
  unsigned w, h;
  uint8_t *score;
  gs_keypoint *out; /* host */
  unsigned nkps;    /* wanted */
  unsigned got;
};

void orb_extract_levels(OrbLevel *L, unsigned nl, unsigned threshold) {
  hipStream_t st = ctx().s();
  unsigned cap[4], coff[4], ctot = 0;
  for (unsigned l = 0; l < nl; l++) {
    cap[l] = (L[l].nkps && L[l].w >= 7 && L[l].h >= 7) ? std::min(L[l].nkps * 4u, 5000u) : 0u;
    coff[l] = ctot, ctot += cap[l], L[l].got = 0;
  }
  if (!ctot) return;
  unsigned *kps = (unsigned *)ctx().scratch(SL_KPS, (size_t)ctot * 48 + 16);
  unsigned *cnt = (unsigned *)ctx().scratch(SL_TOT, 16);
  int *mom = (int *)ctx().scratch(SL_MOM, (size_t)ctot * 8);
  for (unsigned l = 0; l < nl; l++) {
    if (!cap[l]) continue;
    launch_fast(L[l].img, L[l].score, L[l].w, L[l].h, 1, kps + (size_t)coff[l] * 12, cnt + l, cap[l], threshold);
    /* moments of every candidate slot (blocks beyond the device-side count exit) */
    GS_LAUNCH(k_orient_moments, dim3(cap[l]), dim3(64), 0, st, L[l].img, L[l].w, L[l].h,
              (const unsigned *)(kps + (size_t)coff[l] * 12), 12u, 15u, mom + (size_t)coff[l] * 2,
              (const unsigned *)(cnt + l));
  }
  std::vector<unsigned> hk((size_t)ctot * 12);
  std::vector<int> hm((size_t)ctot * 2);
  unsigned hn[4] = {0, 0, 0, 0};
  GS_HIP(hipMemcpyAsync(hn, cnt, 16, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(hk.data(), kps, (size_t)ctot * 48, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(hm.data(), mom, (size_t)ctot * 8, hipMemcpyDeviceToHost, st));
  ctx().sync();
  /* host half of ref :657-667 */
  struct Cand { unsigned x, y, response; int m01, m10; };
  std::vector<KpIn> kin;
  unsigned koff[4], ktot = 0;
  const unsigned r = 15;
  for (unsigned l = 0; l < nl; l++) {
    koff[l] = ktot;
    if (!cap[l]) continue;
    const unsigned n = std::min(hn[l], cap[l]);
    std::vector<Cand> cand(n);
    for (unsigned i = 0; i < n; i++) {
      const size_t q = (size_t)coff[l] + i;
      cand[i] = Cand{hk[q * 12], hk[q * 12 + 1], hk[q * 12 + 2], hm[2 * q], hm[2 * q + 1]};
    }
    std::stable_sort(cand.begin(), cand.end(),
                     [](const Cand &a, const Cand &b) { return a.response > b.response; });
    unsigned kept = 0;
    for (size_t i = 0; i < cand.size() && kept < L[l].nkps; i++) {
      const Cand &c = cand[i];
      if (c.x >= r && c.y >= r && c.x < L[l].w - r && c.y < L[l].h - r) {
        gs_keypoint &k = L[l].out[kept];
        k.pt.x = c.x, k.pt.y = c.y, k.response = c.response;
        k.angle = atan2f((float)c.m01, (float)c.m10); /* ref :620, :100 */
        const float angle = k.angle;
        kin.push_back(KpIn{c.x, c.y, sinf(angle), sinf((float)(angle + 1.57079f))}); /* ref :626 */
        kept++;
      }
    }
    L[l].got = kept, ktot += kept;
  }
  if (!ktot) return;
  KpIn *dk = (KpIn *)ctx().scratch(SL_KIN, (size_t)ktot * sizeof(KpIn));
  uint32_t *dd = (uint32_t *)ctx().scratch(SL_DESC, (size_t)ktot * 32);
  GS_HIP(hipMemcpyAsync(dk, kin.data(), (size_t)ktot * sizeof(KpIn), hipMemcpyHostToDevice, st));
  for (unsigned l = 0; l < nl; l++)
    if (L[l].got)
      GS_LAUNCH(k_brief, dim3(L[l].got), dim3(256), 0, st, L[l].img, L[l].w, L[l].h,
                (const KpIn *)(dk + koff[l]), dd + (size_t)koff[l] * 8);
  std::vector<uint32_t> hd((size_t)ktot * 8);
  GS_HIP(hipMemcpyAsync(hd.data(), dd, (size_t)ktot * 32, hipMemcpyDeviceToHost, st));
  ctx().sync();
  for (unsigned l = 0; l < nl; l++)
    for (unsigned i = 0; i < L[l].got; i++)
      memcpy(L[l].out[i].descriptor, &hd[((size_t)koff[l] + i) * 8], 32);
}

void launch_match(const uint32_t *k1, unsigned n1, const uint32_t *k2, unsigned n2,
                  unsigned *matches, unsigned *count, unsigned max_matches, float max_distance) {
  hipStream_t st = ctx().s();
  if (n1 == 0 || max_matches == 0) {
    GS_HIP(hipMemsetAsync(count, 0, 4, st));
    return;
  }
  const unsigned blocks = (n1 + 3) / 4, words = (n1 + 63) / 64;
  const unsigned nchunks = (words + kChunkWords - 1) / kChunkWords;
  unsigned long long *mask =
      (unsigned long long *)ctx().scratch(SL_MASK, (size_t)nchunks * kChunkWords * 8);
  unsigned *cnt = (unsigned *)ctx().scratch(SL_CNT, (size_t)nchunks * 4);
  unsigned *best = (unsigned *)ctx().scratch(SL_BEST, (size_t)n1 * 8);
  GS_HIP(hipMemsetAsync(mask, 0, (size_t)nchunks * kChunkWords * 8, st));
  GS_HIP(hipMemsetAsync(cnt, 0, (size_t)nchunks * 4, st));
  GS_LAUNCH(k_match, dim3(blocks), dim3(256), 0, st, k1, n1, k2, n2, max_distance, best, best + n1,
            mask, cnt);
  run_compaction(mask, cnt, nchunks, 1, max_matches, count,
                 MatchEmit{best, best + n1, matches});
}

void synth_jump_table(SynthJump &J) {
  auto step = [](uint32_t x) {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
  };
  for (int b = 0; b < 32; b++) J.col[0][b] = step(1u << b);
  for (int k = 1; k < 32; k++)
    for (int b = 0; b < 32; b++) {
      uint32_t v = J.col[k - 1][b], r = 0;
      for (int q = 0; q < 32; q++)
        if ((v >> q) & 1u) r ^= J.col[k - 1][q];
      J.col[k][b] = r;
    }
}

}  // namespace

/* =====================================================================================
 *                                   C ABI
 * ===================================================================================== */
extern "C" {

const char *gsh_version(void) {
#ifdef GS_EMU
  return "grayskull_hip 0.1 (kernel-logic emulator build -- test tool, not a product)";
#else
  return "grayskull_hip 0.1 gfx950";
#endif
}
int gsh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
void gsh_set_device(int ordinal) {
  Ctx &c = ctx();
  if (c.device_set && c.device != ordinal) c.release();
  c.device = ordinal;
  c.device_set = false;
  c.ensure_device();
}
void gsh_set_stream(void *s) {
  Ctx &c = ctx();
  if (c.own_stream) {
    c.sync();
    (void)hipStreamDestroy(c.stream);
    c.own_stream = false;
  }
  c.stream = (hipStream_t)s;
  c.user_stream = s != nullptr;
}
void *gsh_get_stream(void) { return (void *)ctx().s(); }
void gsh_set_async(int on) { ctx().async = on != 0; }
void gsh_profile(int on) {
#ifndef GS_EMU
  Ctx &c = ctx();
  c.ensure_device();
  c.prof_on = on != 0;
  c.prof_n = 0;
  /* on > 1: create that many event pairs now, so that none is created inside a timed region */
  for (int i = 0; on > 1 && i < 2 * std::min(on, (int)Ctx::kProfPairs); i++)
    if (!c.prof_ev[i]) GS_HIP(hipEventCreateWithFlags(&c.prof_ev[i], sync_event_flags())); /* timing on, no system fence */
#else
  (void)on;
#endif
}
unsigned gsh_profile_read(double *total_ms) {
  unsigned n = 0;
  double sum = 0;
#ifndef GS_EMU
  Ctx &c = ctx();
  c.sync();
  for (unsigned i = 0; i < c.prof_n; i++) {
    float ms = 0;
    GS_HIP(hipEventSynchronize(c.prof_ev[2 * i + 1]));
    GS_HIP(hipEventElapsedTime(&ms, c.prof_ev[2 * i], c.prof_ev[2 * i + 1]));
    sum += ms;
  }
  n = c.prof_n;
  c.prof_n = 0;
#endif
  if (total_ms) *total_ms = sum;
  return n;
}
void gsh_tune(int key, int value) {
  if (key >= 0 && key < 32) g_tune.set(key, value);
}
void gsh_probe_strip_copy(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w >= 32);
  const StripCfg c = strip_cfg(w, h, n, 5, 2, g_tune[0] > 0 ? (unsigned)g_tune[0] : 8u, ragged(w) ? 1 : 0);
  if (g_tune[23] == 1) GS_LAUNCH((k_strip_copy<0, true>), c.grid, c.block, 0, ctx().s(), dst, src, w, h, c.T, (size_t)w * h | c.xcd_flag);
  else if (ragged(w)) GS_LAUNCH(k_strip_copy<1>, c.grid, c.block, 0, ctx().s(), dst, src, w, h, c.T, (size_t)w * h | c.xcd_flag);
  else GS_LAUNCH(k_strip_copy<0>, c.grid, c.block, 0, ctx().s(), dst, src, w, h, c.T, (size_t)w * h | c.xcd_flag);
}
void gsh_probe_fast_score(uint8_t *score, const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned threshold) {
  GS_ASSERT(score && img && w >= 7 && h >= 7 && n >= 1 && n <= kMaxZ);
  launch_fast_score(ctx().s(), img, score, w, h, n, threshold);
}
void gsh_sync(void) { ctx().sync(); }
void gsh_shutdown(void) { ctx().release(); }
void *gsh_malloc(size_t bytes) {
  ctx().ensure_device();
  void *p = nullptr;
  GS_HIP(hipMalloc(&p, bytes ? bytes : 1));
  return p;
}
void gsh_free(void *p) {
  if (p) GS_HIP(hipFree(p));
}
void *gsh_host_alloc(size_t bytes) {
  ctx().ensure_device();
  void *p = nullptr;
  GS_HIP(hipHostMalloc(&p, bytes ? bytes : 1, 0));
  return p;
}
void gsh_host_free(void *p) {
  if (p) GS_HIP(hipHostFree(p));
}
void gsh_memset(void *dev, int byte, size_t bytes) {
  GS_HIP(hipMemsetAsync(dev, byte, bytes, ctx().s()));
}
void gsh_upload(void *dev, const void *host, size_t bytes) {
  GS_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx().s()));
  ctx().sync();
}
void gsh_download(void *host, const void *dev, size_t bytes) {
  GS_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
}
int gsh_is_device_ptr(const void *p) { return is_dev(p) ? 1 : 0; }

/* ---------------------------------------------------------------- batch: stencils */
void gsh_blur_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                    unsigned radius) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_blur(dst, src, w, h, n, radius);
}
void gsh_sobel_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_sobel(dst, src, w, h, n);
}
void gsh_erode_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_morph<false>(dst, src, w, h, n);
}
void gsh_dilate_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  launch_morph<true>(dst, src, w, h, n);
}

/* ---------------------------------------------------------------- batch: histogram etc. */
void gsh_histogram_batch(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist) {
  GS_ASSERT(img && hist && w > 0 && h > 0);
  launch_histogram(img, (size_t)(w * h), n, hist); /* 32-bit product like ref :202 */
}
void gsh_otsu_batch(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist_scratch,
                    uint8_t *thr) {
  GS_ASSERT(img && hist_scratch && thr && w > 0 && h > 0);
  launch_otsu(img, w, h, n, hist_scratch, thr);
}
void gsh_threshold_batch(uint8_t *img, unsigned w, unsigned h, unsigned n, uint8_t thresh) {
  GS_ASSERT(img && w > 0 && h > 0);
  launch_threshold(img, (size_t)(w * h), n, nullptr, thresh);
}
void gsh_threshold_batch_dev(uint8_t *img, unsigned w, unsigned h, unsigned n, const uint8_t *thr) {
  GS_ASSERT(img && thr && w > 0 && h > 0);
  launch_threshold(img, (size_t)(w * h), n, thr, 0);
}
/* gs_blur(radius) then gs_sobel into a zeroed image, per frame, in one pass (the fused kernel of
 * the pipeline without the Otsu / threshold half) */
void gsh_blur_sobel_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                          unsigned radius) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  if (n == 0) return;
  const size_t fb = (size_t)w * h;
  hipStream_t st = ctx().s();
  if (g_tune[3] == 0 && radius >= 1 && radius <= 3 && strip_ok(w, h, dst, src) && w >= 32 && h >= 3 &&
      h > 2 * radius) {
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      const int rg = strip_mode(w, src + fb * f0);
      const StripCfg c = strip_cfg(w, h - 2, nn, 3, 2, 8, rg);
      launch_blur_sobel(radius, dim3(c.grid.x, c.grid.y, nn), c.block, st, dst + fb * f0, src + fb * f0, w, h,
                        c.T, fb, rg);
      GS_LAUNCH(k_zero_frame, dim3((2 * w + 2 * h + 255) / 256, nn), dim3(256), 0, st, dst + fb * f0, w, h, fb);
    }
    return;
  }
  uint8_t *t = (uint8_t *)ctx().scratch(SL_AUX, fb * n);
  launch_blur(t, src, w, h, n, radius);
  GS_HIP(hipMemsetAsync(dst, 0, fb * n, st));
  if (w >= 3 && h >= 3) launch_sobel(dst, t, w, h, n, true);
}
void gsh_edge_pipeline_batch(uint8_t *dst, uint8_t *tmp, const uint8_t *src, unsigned w, unsigned h,
                             unsigned n, unsigned radius, unsigned *hist_scratch, uint8_t *thr) {
  GS_ASSERT(dst && src && hist_scratch && thr && w > 0 && h > 0);
  const size_t fb = (size_t)w * h;
  hipStream_t st = ctx().s();
  auto zero_frame = [&]() { /* gs_sobel ran "into a zeroed image": only its 1-px frame is left */
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ)
      GS_LAUNCH(k_zero_frame, dim3((2 * w + 2 * h + 255) / 256, std::min(kMaxZ, n - f0)), dim3(256), 0,
                st, dst + fb * f0, w, h, fb);
  };
  if (!tmp && g_tune[3] == 0 && radius >= 1 && radius <= 3 && strip_ok16(w, h, dst, src) && w >= 32 &&
      h >= 3 && h > 2 * radius) { /* every window is clipped on at most one side per axis */
    /* fused: the blurred image only ever exists in registers (1 R + 1 W per pixel).
     * The fused kernel is VALU-bound (~40 % of HBM peak) and the passes after it HBM-bound with
     * an idle VALU.  A batch larger than kChunkFrames is cut into chunks: the fused kernels run
     * back to back on the caller's stream, each chunk's histogram reduce / Otsu / frame zeroing /
     * threshold pass on a side stream behind one event, i.e. under the next chunk's fused kernel.
     * gsh_tune key 5: frames per chunk (0 = default, negative = never split). */
    const int tune_chunk = g_tune[5];
    const unsigned per = tune_chunk < 0 ? n : tune_chunk > 0 ? (unsigned)tune_chunk : kChunkFrames;
    /* chunk sizes: `per` frames each.  (Tapering the tail -- 16, 8, 8 -- so that less of the last
     * threshold pass is exposed measured slower: small fused launches cost more than they hide.) */
    std::vector<unsigned> sizes;
    for (unsigned rem = n; rem; rem -= std::min(rem, per)) sizes.push_back(std::min(rem, per));
    /* band height per chunk size (a smaller chunk needs more bands to fill the chip) */
    auto cfg_for = [&](unsigned nn) { return strip_cfg(w, h - 2, std::min(kMaxZ, nn), 3); /* <= 168 VGPRs: 3 waves per SIMD */ };
    unsigned bpf_max = 0;
    for (unsigned nn : sizes) {
      const StripCfg c = cfg_for(nn);
      bpf_max = std::max(bpf_max, c.grid.x * c.grid.y);
    }
    unsigned *partial = (unsigned *)ctx().scratch(SL_HISTP, (size_t)n * bpf_max * 256 * 4);
    auto run_fused = [&](hipStream_t on, unsigned f0, unsigned nn) {
      const StripCfg c = cfg_for(nn);
      for (unsigned g0 = f0; g0 < f0 + nn; g0 += kMaxZ) {
        const unsigned m = std::min(kMaxZ, f0 + nn - g0);
        const dim3 grid(c.grid.x, c.grid.y, m);
        uint8_t *d = dst + fb * g0;
        const uint8_t *sp = src + fb * g0;
        unsigned *pp = partial + (size_t)g0 * bpf_max * 256;
#ifndef GS_EMU
        ctx().prof_mark(0, on);
#endif
        launch_blur_sobel_hist(radius, grid, c.block, on, d, sp, w, h, c.T, fb, pp);
#ifndef GS_EMU
        ctx().prof_mark(1, on);
#endif
      }
    };
    auto run_rest = [&](hipStream_t on, unsigned f0, unsigned nn) {
      const StripCfg c = cfg_for(nn);
      const unsigned bpf = c.grid.x * c.grid.y;
      for (unsigned g0 = f0; g0 < f0 + nn; g0 += kMaxZ) {
        const unsigned m = std::min(kMaxZ, f0 + nn - g0);
        const unsigned *pp = partial + (size_t)g0 * bpf_max * 256;
        /* the 2w + 2(h-2) frame pixels are 0 in the result and were not counted by the kernel; the threshold pass
         * below waits for this launch only (k_otsu folds the blocks' partial histograms itself) */
        GS_LAUNCH(k_otsu, dim3(m), dim3(256), 0, on, hist_scratch + (size_t)g0 * 256, w * h, thr + g0, pp, bpf,
                  2 * w + 2 * (h - 2));
      }
      launch_threshold(dst + fb * f0, fb, nn, thr + f0, 0, on);
      /* gs_sobel ran "into a zeroed image": only its 1-px frame is left to zero.  AFTER the threshold pass (which
       * turns whatever the fused kernel left there into 0 / 255): the frame is then 0 either way, and the
       * threshold pass need not wait for this launch. */
      for (unsigned g0 = f0; g0 < f0 + nn; g0 += kMaxZ) {
        const unsigned m = std::min(kMaxZ, f0 + nn - g0);
        GS_LAUNCH(k_zero_frame, dim3((2 * w + 2 * h + 255) / 256, m), dim3(256), 0, on, dst + fb * g0, w, h, fb);
      }
    };
#ifdef GS_EMU
    const bool split = false;
#else
    const bool split = sizes.size() > 1 && sizes.size() <= kMaxChunks;
#endif
    if (!split) {
      run_fused(st, 0, n);
      run_rest(st, 0, n);
      return;
    }
#ifndef GS_EMU
    Ctx &cx = ctx();
    cx.ensure_side();
    unsigned f0 = 0;
    for (size_t i = 0; i < sizes.size(); i++) {
      const unsigned nn = sizes[i];
      run_fused(st, f0, nn);
      if (i + 1 == sizes.size()) { /* last chunk: nothing left to hide it under; rejoin the caller's stream */
        GS_HIP(hipEventRecord(cx.ev_join, cx.side));
        GS_HIP(hipStreamWaitEvent(st, cx.ev_join, 0));
        run_rest(st, f0, nn);
      } else {
        GS_HIP(hipEventRecord(cx.ev_chunk[i], st));
        GS_HIP(hipStreamWaitEvent(cx.side, cx.ev_chunk[i], 0));
        run_rest(cx.side, f0, nn);
      }
      f0 += nn;
    }
#endif
    return;
  }
  if (!tmp && g_tune[3] == 0 && radius >= 1 && radius <= 3 && strip_ok(w, h, dst, src) && w >= 32 && h >= 3 &&
      h > 2 * radius) {
    /* ragged rows or frames at odd addresses: blur + sobel still in one pass (the fused kernel without its histogram
     * half), the histogram as a pass of its own: 5 B/px moved instead of 4, against 9 for the separate calls */
    gsh_blur_sobel_batch(dst, src, w, h, n, radius);
    launch_otsu(dst, w, h, n, hist_scratch, thr);
    launch_threshold(dst, fb, n, thr, 0);
    return;
  }
  uint8_t *t = tmp ? tmp : (uint8_t *)ctx().scratch(SL_AUX, fb * n);
  launch_blur(t, src, w, h, n, radius);
  /* sobel never writes its 1-px frame (ref :308-309); config 2 runs it into a zeroed image, so
   * only that frame needs zeroing (the interior is overwritten) -- after the sobel launch,
   * which then need not preserve columns 0 / w-1 */
  if (w < 3 || h < 3) GS_HIP(hipMemsetAsync(dst, 0, fb * n, st));
  launch_sobel(dst, t, w, h, n, false);
  if (w >= 3 && h >= 3) zero_frame();
  launch_otsu(dst, w, h, n, hist_scratch, thr);
  launch_threshold(dst, fb, n, thr, 0);
}

/* ---------------------------------------------------------------- batch: integral + LBP */
void gsh_integral_batch(const uint8_t *src, unsigned w, unsigned h, unsigned n, unsigned *ii) {
  GS_ASSERT(src && ii && w > 0 && h > 0);
  launch_integral(src, w, h, n, ii);
}

gsh_cascade *gsh_cascade_create(const struct gs_lbp_cascade *c) {
  GS_ASSERT(c && c->features && c->weak_feature_idx && c->subsets);
  ctx().ensure_device();
  gsh_cascade *dc = new gsh_cascade();
  static std::atomic<unsigned long long> next_id{1};
  dc->id = next_id.fetch_add(1);
  dc->window_w = c->window_w, dc->window_h = c->window_h;
  dc->nfeatures = c->nfeatures, dc->nweaks = c->nweaks, dc->nstages = c->nstages;
  dc->features.assign(c->features, c->features + (size_t)c->nfeatures * 4);
  dc->weak_feature_idx.assign(c->weak_feature_idx, c->weak_feature_idx + c->nweaks);
  /* device tables in EVALUATION order: stage by stage, so a stage range is one contiguous run of
   * weak classifiers (the reference indexes through stage_weak_start, ref :795-798) */
  std::vector<LbpWeak> wk;
  std::vector<LbpStage> stg(c->nstages);
  dc->weak_feature_idx.clear();
  unsigned nsub = 0;
  for (unsigned si = 0; si < c->nstages; si++) {
    stg[si] = LbpStage{(unsigned)wk.size(), c->stage_nweaks[si], c->stage_threshold[si], 0.0f};
    for (unsigned k = 0; k < c->stage_nweaks[si]; k++) {
      const unsigned i = (unsigned)c->stage_weak_start[si] + k;
      wk.push_back(LbpWeak{c->weak_left_val[i], c->weak_right_val[i], c->weak_subset_offset[i],
                           c->weak_num_subsets[i]});
      dc->weak_feature_idx.push_back(c->weak_feature_idx[i]);
      nsub = std::max(nsub, (unsigned)c->weak_subset_offset[i] + c->weak_num_subsets[i]);
    }
  }
  dc->nweaks = (unsigned)wk.size(); /* classifiers actually reachable through the stages */
  dc->nsub = nsub;
  GS_HIP(hipMalloc((void **)&dc->d_weak, std::max<size_t>(1, wk.size()) * sizeof(LbpWeak)));
  GS_HIP(hipMalloc((void **)&dc->d_stage, std::max<size_t>(1, stg.size()) * sizeof(LbpStage)));
  GS_HIP(hipMalloc((void **)&dc->d_subsets, std::max<size_t>(1, nsub) * 4));
  GS_HIP(hipMemcpy(dc->d_weak, wk.data(), wk.size() * sizeof(LbpWeak), hipMemcpyHostToDevice));
  GS_HIP(hipMemcpy(dc->d_stage, stg.data(), stg.size() * sizeof(LbpStage), hipMemcpyHostToDevice));
  GS_HIP(hipMemcpy(dc->d_subsets, c->subsets, (size_t)nsub * 4, hipMemcpyHostToDevice));
  /* Truth tables of the stage decisions (k_lbp_dense.h): for every combination b of a stage's match bits the
   * reference's own sequence -- sum = 0.0f; sum += match ? left : right in weak order; pass unless
   * sum < threshold (ref :796-810) -- evaluated here in float32 (this file is built -ffp-contract=off). */
  std::vector<unsigned> lut(std::max<size_t>(1, stg.size()), 0u);
  dc->pre_max = 0;
  bool leading = true;
  for (unsigned si = 0; si < c->nstages; si++) {
    if (stg[si].count > kPreMaxWeaks) {
      leading = false;
      continue;
    }
    for (unsigned b = 0; b < (1u << stg[si].count); b++) {
      volatile float sum = 0.0f;
      for (unsigned k = 0; k < stg[si].count; k++) {
        const LbpWeak &w = wk[stg[si].first + k];
        sum = sum + (((b >> k) & 1u) ? w.left : w.right);
      }
      if (!(sum < stg[si].threshold)) lut[si] |= 1u << b;
    }
    if (leading) dc->pre_max = si + 1;
  }
  GS_HIP(hipMalloc((void **)&dc->d_pass_lut, lut.size() * 4));
  GS_HIP(hipMemcpy(dc->d_pass_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice));
  return dc;
}
void gsh_cascade_destroy(gsh_cascade *dc) {
  if (!dc) return;
  ctx().sync();
  /* geometry tables of this handle in the calling thread's cache go with it; other threads' entries
   * are keyed by the (never reused) id and are dropped when their context is released */
  auto it = ctx().geom_cache.find(dc->id);
  if (it != ctx().geom_cache.end()) {
    if (it->second.d_scales) (void)hipFree(it->second.d_scales);
    if (it->second.d_geom) (void)hipFree(it->second.d_geom);
    if (it->second.d_pre) (void)hipFree(it->second.d_pre);
    ctx().geom_cache.erase(it);
  }
  gsh_cascade_tables_deleter()(dc);
}
void gsh_lbp_detect_batch(const gsh_cascade *dc, const unsigned *ii, unsigned iw, unsigned ih,
                          unsigned n, struct gs_rect *rects, unsigned *counts, unsigned max_rects,
                          float scale_factor, float min_scale, float max_scale, int step) {
  GS_ASSERT(dc && ii && rects && counts && iw > 0 && ih > 0);
  launch_lbp_unpadded(dc, ii, iw, ih, n, (unsigned *)rects, counts,
                      max_rects, scale_factor, min_scale, max_scale, step);
}
void gsh_lbp_count_evaluated(unsigned long long *counter_dev) { g_lbp_evaluated = counter_dev; }
uint64_t gsh_lbp_window_count(const struct gs_lbp_cascade *c, unsigned iw, unsigned ih,
                              float scale_factor, float min_scale, float max_scale, int step) {
  gsh_cascade tmp;
  tmp.window_w = c->window_w, tmp.window_h = c->window_h, tmp.nweaks = 0;
  std::vector<LbpScale> sc;
  std::vector<LbpGeom> ge;
  bool guard;
  unsigned long long nwin;
  build_scales(tmp, iw, ih, scale_factor, min_scale, max_scale, step, sc, ge, guard, nwin);
  return nwin;
}

/* ---------------------------------------------------------------- batch: FAST / ORB / match */
void gsh_fast_batch(const uint8_t *img, uint8_t *scoremap, unsigned w, unsigned h, unsigned n,
                    struct gs_keypoint *kps, unsigned *counts, unsigned nkps, unsigned threshold) {
  GS_ASSERT(img && scoremap && kps && counts && nkps > 0 && w > 0 && h > 0);
  launch_fast(img, scoremap, w, h, n, (unsigned *)kps, counts, nkps, threshold);
}
unsigned gsh_orb_extract(const uint8_t *img_dev, unsigned w, unsigned h, uint8_t *scoremap_dev,
                         struct gs_keypoint *kps_host, unsigned nkps, unsigned threshold) {
  GS_ASSERT(img_dev && scoremap_dev && kps_host && nkps > 0 && w > 0 && h > 0);
  OrbLevel L{img_dev, w, h, scoremap_dev, kps_host, nkps, 0};
  orb_extract_levels(&L, 1, threshold);
  return L.got;
}
/* gs_orb_extract (ref :651-669) for n frames of one size with two host round trips in total:
 * FAST + NMS + emit and the disc moments run as batch launches over all frames. */
void gsh_orb_extract_batch(const uint8_t *img_dev, unsigned w, unsigned h, unsigned n,
                           uint8_t *scoremap_dev, struct gs_keypoint *kps_host, unsigned *counts_host,
                           unsigned nkps, unsigned threshold) {
  GS_ASSERT(img_dev && scoremap_dev && kps_host && counts_host && nkps > 0 && w > 0 && h > 0);
  constexpr unsigned kOrbGroup = 4096; /* frames per pass: bounds the host-side candidate buffers and grid.y */
  if (n > kOrbGroup) {
    for (unsigned f0 = 0; f0 < n; f0 += kOrbGroup)
      gsh_orb_extract_batch(img_dev + (size_t)w * h * f0, w, h, std::min(kOrbGroup, n - f0),
                            scoremap_dev + (size_t)w * h * f0, kps_host + (size_t)f0 * nkps, counts_host + f0, nkps,
                            threshold);
    return;
  }
  for (unsigned f = 0; f < n; f++) counts_host[f] = 0;
  if (n == 0 || w < 7 || h < 7) return;
  hipStream_t st = ctx().s();
  const size_t fb = (size_t)w * h;
  const unsigned cap = std::min(nkps * 4u, 5000u), r = 15;
  unsigned *kps = (unsigned *)ctx().scratch(SL_KPS, (size_t)n * cap * 48 + 16);
  unsigned *cnt = (unsigned *)ctx().scratch(SL_TOT, (size_t)n * 4 + 16);
  int *mom = (int *)ctx().scratch(SL_MOM, (size_t)n * cap * 8);
  launch_fast(img_dev, scoremap_dev, w, h, n, kps, cnt, cap, threshold);
  GS_LAUNCH(k_orient_moments, dim3(cap, n), dim3(64), 0, st, img_dev, w, h, (const unsigned *)kps, 12u, r, mom,
            (const unsigned *)cnt, fb);
  std::vector<unsigned> hk((size_t)n * cap * 12), hn(n);
  std::vector<int> hm((size_t)n * cap * 2);
  GS_HIP(hipMemcpyAsync(hn.data(), cnt, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(hk.data(), kps, hk.size() * 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(hm.data(), mom, hm.size() * 4, hipMemcpyDeviceToHost, st));
  ctx().sync();
  struct Cand { unsigned x, y, response; int m01, m10; };
  std::vector<KpIn> kin;
  std::vector<unsigned> koff(n);
  std::vector<Cand> cand;
  for (unsigned f = 0; f < n; f++) { /* host half of ref :657-667, frame by frame */
    koff[f] = (unsigned)kin.size();
    const unsigned m = std::min(hn[f], cap);
    cand.resize(m);
    for (unsigned i = 0; i < m; i++) {
      const size_t q = (size_t)f * cap + i;
      cand[i] = Cand{hk[q * 12], hk[q * 12 + 1], hk[q * 12 + 2], hm[2 * q], hm[2 * q + 1]};
    }
    std::stable_sort(cand.begin(), cand.end(),
                     [](const Cand &a, const Cand &b) { return a.response > b.response; });
    unsigned kept = 0;
    gs_keypoint *out = kps_host + (size_t)f * nkps;
    for (size_t i = 0; i < cand.size() && kept < nkps; i++) {
      const Cand &c = cand[i];
      if (c.x >= r && c.y >= r && c.x < w - r && c.y < h - r) {
        gs_keypoint &k = out[kept];
        k.pt.x = c.x, k.pt.y = c.y, k.response = c.response;
        k.angle = atan2f((float)c.m01, (float)c.m10); /* ref :620, :100 */
        const float angle = k.angle;
        kin.push_back(KpIn{c.x, c.y, sinf(angle), sinf((float)(angle + 1.57079f))}); /* ref :626 */
        kept++;
      }
    }
    counts_host[f] = kept;
  }
  if (kin.empty()) return;
  KpIn *dk = (KpIn *)ctx().scratch(SL_KIN, kin.size() * sizeof(KpIn));
  uint32_t *dd = (uint32_t *)ctx().scratch(SL_DESC, kin.size() * 32);
  GS_HIP(hipMemcpyAsync(dk, kin.data(), kin.size() * sizeof(KpIn), hipMemcpyHostToDevice, st));
  for (unsigned f = 0; f < n; f++)
    if (counts_host[f])
      GS_LAUNCH(k_brief, dim3(counts_host[f]), dim3(256), 0, st, img_dev + fb * f, w, h,
                (const KpIn *)(dk + koff[f]), dd + (size_t)koff[f] * 8);
  std::vector<uint32_t> hd(kin.size() * 8);
  GS_HIP(hipMemcpyAsync(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost, st));
  ctx().sync();
  for (unsigned f = 0; f < n; f++)
    for (unsigned i = 0; i < counts_host[f]; i++)
      memcpy(kps_host[(size_t)f * nkps + i].descriptor, &hd[((size_t)koff[f] + i) * 8], 32);
}

/* gs_orb_extract (ref :651-669) for n frames, everything on the device, no host round trip: FAST ->
 * selection (stable descending sort + 15-px border filter + cap, k_orb_select) -> orientation + BRIEF
 * (k_orb_describe) with the reference's GS_NO_STDLIB trig (ref :70-88).  Results are those of the
 * reference header compiled with -DGS_NO_STDLIB; the libm flavour (glibc atan2f / sinf bits) stays
 * with gsh_orb_extract_batch, whose trig runs on the host. */
void gsh_orb_extract_batch_nostdlib(const uint8_t *img_dev, unsigned w, unsigned h, unsigned n,
                                    uint8_t *scoremap_dev, struct gs_keypoint *kps_dev, unsigned *counts_dev,
                                    unsigned nkps, unsigned threshold) {
  GS_ASSERT(img_dev && scoremap_dev && kps_dev && counts_dev && nkps > 0 && w > 0 && h > 0);
  if (n == 0) return;
  hipStream_t st = ctx().s();
  if (w < 7 || h < 7) {
    GS_HIP(hipMemsetAsync(counts_dev, 0, (size_t)n * 4, st));
    return;
  }
  const size_t fb = (size_t)w * h;
  const unsigned cap = std::min(nkps * 4u, 5000u);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    unsigned *cand = (unsigned *)ctx().scratch(SL_KPS, (size_t)nn * cap * 48 + 16);
    unsigned *cnt = (unsigned *)ctx().scratch(SL_TOT, (size_t)nn * 4 + 16);
    launch_fast(img_dev + fb * f0, scoremap_dev + fb * f0, w, h, nn, cand, cnt, cap, threshold);
    unsigned *out = (unsigned *)kps_dev + (size_t)f0 * nkps * 12;
    GS_LAUNCH(k_orb_select, dim3(nn), dim3(64), 0, st, (const unsigned *)cand, (const unsigned *)cnt, cap, w, h, nkps,
              out, counts_dev + f0);
    GS_LAUNCH(k_orb_describe, dim3(nkps, nn), dim3(256), 0, st, img_dev + fb * f0, w, h, fb, out,
              (const unsigned *)(counts_dev + f0), nkps);
  }
}

size_t gsh_orb_pyramid_buffer_bytes(unsigned w, unsigned h, unsigned n_levels) {
  if (n_levels > 4) n_levels = 4;
  size_t levels = 0, maps = (size_t)w * h;
  for (unsigned l = 1; l < n_levels; l++) {
    w /= 2, h /= 2;
    if (w < 32 || h < 32) break;
    levels += (size_t)w * h, maps += (size_t)w * h;
  }
  return levels + maps;
}
/* ref examples/nanomagick/nanomagick.c:245-290 with every level resident on the device */
unsigned gsh_orb_extract_pyramid(const uint8_t *img_dev, unsigned w, unsigned h, uint8_t *buffer_dev,
                                 struct gs_keypoint *kps_host, unsigned nkps, unsigned threshold,
                                 unsigned n_levels) {
  GS_ASSERT(img_dev && buffer_dev && kps_host && w > 0 && h > 0);
  if (n_levels > 4) n_levels = 4;
  if (n_levels == 0 || nkps == 0) return 0; /* the reference driver's level loop is empty (nanomagick.c:262) */
  const uint8_t *lev[4];
  unsigned lw[4], lh[4], total = 0;
  size_t off = 0;
  lev[0] = img_dev, lw[0] = w, lh[0] = h;
  for (unsigned l = 1; l < n_levels; l++) {
    const unsigned nw = lw[l - 1] / 2, nh = lh[l - 1] / 2;
    if (nw < 32 || nh < 32) {
      n_levels = l;
      break;
    }
    uint8_t *d = buffer_dev + off;
    off += (size_t)nw * nh;
    gsh_downsample_batch(d, lev[l - 1], lw[l - 1], lh[l - 1], 1);
    lev[l] = d, lw[l] = nw, lh[l] = nh;
  }
  /* per-level quotas depend on how many keypoints the earlier levels produced only for the last
   * level ("the remainder", nanomagick.c:275): run the first n_levels-1 levels as one batch, then
   * the last one */
  OrbLevel L[4];
  uint8_t *sm[4];
  for (unsigned l = 0; l < n_levels; l++) {
    sm[l] = buffer_dev + off;
    off += (size_t)lw[l] * lh[l];
  }
  const unsigned per = nkps / n_levels;
  for (unsigned l = 0; l + 1 < n_levels; l++) L[l] = OrbLevel{lev[l], lw[l], lh[l], sm[l], kps_host + (size_t)l * per, per, 0};
  if (n_levels > 1) orb_extract_levels(L, n_levels - 1, threshold);
  for (unsigned l = 0; l + 1 < n_levels; l++) {
    /* levels write at l*per; the reference packs them back to back (total_kps) */
    if (L[l].got && total != l * per) memmove(kps_host + total, kps_host + (size_t)l * per, (size_t)L[l].got * sizeof(gs_keypoint));
    for (unsigned i = total; i < total + L[l].got; i++) kps_host[i].pt.x <<= l, kps_host[i].pt.y <<= l;
    total += L[l].got;
  }
  {
    const unsigned l = n_levels - 1, want = nkps - total;
    if (want) {
      L[l] = OrbLevel{lev[l], lw[l], lh[l], sm[l], kps_host + total, want, 0};
      orb_extract_levels(&L[l], 1, threshold);
      for (unsigned i = total; i < total + L[l].got; i++) kps_host[i].pt.x <<= l, kps_host[i].pt.y <<= l;
      total += L[l].got;
    }
  }
  return total;
}
void gsh_match_orb_dev(const struct gs_keypoint *k1, unsigned n1, const struct gs_keypoint *k2,
                       unsigned n2, struct gs_match *matches, unsigned *count,
                       unsigned max_matches, float max_distance) {
  GS_ASSERT(k1 && k2 && matches && count);
  launch_match((const uint32_t *)k1, n1, (const uint32_t *)k2, n2, (unsigned *)matches, count,
               max_matches, max_distance);
}

/* ---------------------------------------------------------------- batch: "next" rows */
void gsh_adaptive_threshold_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                                  unsigned n, unsigned radius, int c) {
  GS_ASSERT(dst && src && w > 0 && h > 0);
  if (n) launch_box_generic<1>(dst, src, w, h, n, radius, c);
}
void gsh_filter_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                      const int8_t *kernel_host, unsigned kw, unsigned kh, unsigned norm) {
  GS_ASSERT(dst && src && kernel_host && w > 0 && h > 0 && kw > 0 && kh > 0 && norm > 0);
  hipStream_t st = ctx().s();
  int8_t *dk = (int8_t *)ctx().scratch(SL_TAB, (size_t)kw * kh);
  GS_HIP(hipMemcpyAsync(dk, kernel_host, (size_t)kw * kh, hipMemcpyHostToDevice, st));
  ctx().sync(); /* kernel_host may be a temporary */
  const size_t fb = (size_t)w * h;
  /* strip kernel: 3x3, every partial sum within int16 (sum |k| <= 128), norm <= 256 */
  unsigned abs_sum = 0;
  for (unsigned i = 0; i < kw * kh; i++) abs_sum += (unsigned)std::abs((int)kernel_host[i]);
  if (kw == 3 && kh == 3 && abs_sum <= 128 && norm <= 256 && strip_ok(w, h, dst, src) && w >= 32) {
    FilterK fk;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) fk.k[r][c] = ((uint32_t)(uint16_t)(int16_t)kernel_host[r * 3 + c]) * 0x10001u;
    fk.mul = (0x1000000u + norm - 1u) / norm;
    fk.cap = std::min(255u * norm, 32767u) * 0x10001u;
    fk.neg_is_255 = norm > 1 ? 0xffffffffu : 0u;
    for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
      const unsigned nn = std::min(kMaxZ, n - f0);
      const int rg = strip_mode(w, src + fb * f0);
      const StripCfg c = strip_cfg(w, h, nn, 6, 2, 8, rg);
      if (rg == 1) GS_LAUNCH(k_filter16<1>, c.grid, c.block, 0, st, dst + fb * f0, src + fb * f0, w, h, c.T, fb | c.xcd_flag, fk);
      else if (rg == 2) GS_LAUNCH(k_filter16<2>, c.grid, c.block, 0, st, dst + fb * f0, src + fb * f0, w, h, c.T, fb | c.xcd_flag, fk);
      else GS_LAUNCH(k_filter16<0>, c.grid, c.block, 0, st, dst + fb * f0, src + fb * f0, w, h, c.T, fb | c.xcd_flag, fk);
    }
    return;
  }
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_filter_px, grid2d(w, h, nn), dim3(64, 4), 0, st, dst + fb * f0, src + fb * f0, w, h,
              fb, (const int8_t *)dk, kw, kh, norm);
  }
}
void gsh_downsample_batch(uint8_t *dst, const uint8_t *src, unsigned sw, unsigned sh, unsigned n) {
  GS_ASSERT(dst && src && sw > 1 && sh > 1);
  hipStream_t st = ctx().s();
  const size_t sfb = (size_t)sw * sh, dfb = (size_t)(sw / 2) * (sh / 2);
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    if (g_tune[21] == 1 ? (sw % 16 == 0 && al16(src) && al16(dst) && sfb % 16 == 0 && dfb % 8 == 0) : sw >= 16)
      GS_LAUNCH(k_downsample8, dim3(((sw / 2 + 7) / 8 + 63) / 64, (sh / 2 + 3) / 4, nn), dim3(64, 4), 0, st,
                dst + dfb * f0, src + sfb * f0, sw, sh);
    else
      GS_LAUNCH(k_downsample_px, grid2d(sw / 2, sh / 2, nn), dim3(64, 4), 0, st, dst + dfb * f0,
                src + sfb * f0, sw, sh);
  }
}

/* ---------------------------------------------------------------- synthetic frames, checksums */
void gsh_synth_batch(uint8_t *dst, unsigned w, unsigned h, unsigned n, uint32_t seed0) {
  GS_ASSERT(dst && w > 0 && h > 0);
  if (!n) return;
  hipStream_t st = ctx().s();
  bool &jump_ready = ctx().jump_ready; /* cleared when the scratch is released (shutdown / device switch) */
  SynthJump *dj = (SynthJump *)ctx().scratch(SL_JUMP, sizeof(SynthJump));
  if (!jump_ready) {
    SynthJump J;
    synth_jump_table(J);
    GS_HIP(hipMemcpyAsync(dj, &J, sizeof J, hipMemcpyHostToDevice, st));
    ctx().sync();
    jump_ready = true;
  }
  const unsigned nlev = ((w + 31) / 32) * ((h + 31) / 32);
  const size_t npx = (size_t)w * h;
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    uint8_t *lev = (uint8_t *)ctx().scratch(SL_LEV, (size_t)nlev * nn);
    const unsigned per = 256 * kSynthRun;
    GS_LAUNCH(k_synth_levels, dim3((nlev + per - 1) / per, nn), dim3(256), 0, st, lev, nlev,
              seed0 + f0, (const SynthJump *)dj);
    GS_LAUNCH(k_synth_pixels, dim3((unsigned)((npx + per - 1) / per), nn), dim3(256), 0, st,
              dst + npx * f0, (const uint8_t *)lev, w, h, seed0 + f0, (const SynthJump *)dj);
  }
}
void gsh_checksum_batch(const uint8_t *img, size_t frame_bytes, unsigned n, uint64_t *sums) {
  GS_ASSERT(img && sums && frame_bytes > 0);
  if (!n) return;
  hipStream_t st = ctx().s();
  GS_HIP(hipMemsetAsync(sums, 0, (size_t)n * 8, st));
  const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((frame_bytes + 4095) / 4096, 256));
  for (unsigned f0 = 0; f0 < n; f0 += kMaxZ) {
    const unsigned nn = std::min(kMaxZ, n - f0);
    GS_LAUNCH(k_checksum, dim3(bx, nn), dim3(256), 0, st, img + frame_bytes * f0, frame_bytes,
              (unsigned long long *)sums + f0);
  }
}

/* =====================================================================================
 *              drop-in functions: the reference's own names and signatures
 * ===================================================================================== */
#define GS_VALID(i) ((i).data && (i).w > 0 && (i).h > 0)

/* dst written in full by the kernel: upload src, run, download dst */
static void unary_full(struct gs_image dst, struct gs_image src,
                       void (*run)(uint8_t *, const uint8_t *, unsigned, unsigned, unsigned,
                                   unsigned, int),
                       unsigned p0, int p1) {
  const size_t nb = (size_t)src.w * src.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, nb) : dst.data;
  run(d, s, src.w, src.h, 1, p0, p1);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, nb, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

void gs_blur(struct gs_image dst, struct gs_image src, unsigned radius) { /* ref :268 */
  GS_ASSERT(GS_VALID(src) && GS_VALID(dst) && dst.w == src.w && dst.h == src.h);
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned r, int) { launch_blur(d, s, w, h, n, r); }, radius, 0);
}
void gs_erode(struct gs_image dst, struct gs_image src) { /* ref :303 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h);
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned, int) { launch_morph<false>(d, s, w, h, n); }, 0, 0);
}
void gs_dilate(struct gs_image dst, struct gs_image src) { /* ref :304 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h);
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned, int) { launch_morph<true>(d, s, w, h, n); }, 0, 0);
}
void gs_adaptive_threshold(struct gs_image dst, struct gs_image src, unsigned radius, int c) {
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h); /* ref :232 */
  unary_full(dst, src, [](uint8_t *d, const uint8_t *s, unsigned w, unsigned h, unsigned n,
                          unsigned r, int cc) { launch_box_generic<1>(d, s, w, h, n, r, cc); },
             radius, c);
}

void gs_sobel(struct gs_image dst, struct gs_image src) { /* ref :306 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && dst.w == src.w && dst.h == src.h);
  const unsigned w = src.w, h = src.h;
  if (w < 3 || h < 3) return; /* reference loops are empty */
  const size_t nb = (size_t)w * h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, nb) : dst.data;
  if (dhost) {
    /* The 1-px frame of dst is never written (ref :308-309).  Rows 0 / h-1 are simply not copied
     * back; columns 0 / w-1 of the other rows are planted into the device copy (2h bytes up) so
     * rows 1..h-2 can come back as ONE contiguous copy (a pitched interior copy is 3x slower). */
    std::vector<uint8_t> cols(2 * (size_t)h);
    for (unsigned y = 1; y + 1 < h; y++) cols[2 * y] = dst.data[(size_t)y * w], cols[2 * y + 1] = dst.data[(size_t)y * w + w - 1];
    uint8_t *dc = (uint8_t *)ctx().scratch(SL_AUX2, 2 * (size_t)h);
    GS_HIP(hipMemcpyAsync(dc, cols.data(), 2 * (size_t)h, hipMemcpyHostToDevice, ctx().s()));
    GS_LAUNCH(k_put_cols, dim3((2 * h + 255) / 256), dim3(256), 0, ctx().s(), d, (const uint8_t *)dc, w, h);
    ctx().sync(); /* cols is a local */
  }
  launch_sobel(d, s, w, h, 1, true);
  if (dhost)
    GS_HIP(hipMemcpyAsync(dst.data + w, d + w, (size_t)w * (h - 2), hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

void gs_filter(struct gs_image dst, struct gs_image src, struct gs_image kernel, unsigned norm) {
  GS_ASSERT(GS_VALID(src) && GS_VALID(dst) && dst.w == src.w && dst.h == src.h && norm > 0);
  const size_t nb = (size_t)src.w * src.h, kb = (size_t)kernel.w * kernel.h;
  /* ref :260-261: the kernel is read through gs_get, so an invalid kernel contributes nothing */
  std::vector<int8_t> k(std::max<size_t>(1, kb), 0);
  unsigned kw = kernel.w, kh = kernel.h;
  if (GS_VALID(kernel)) {
    if (is_dev(kernel.data)) gsh_download(k.data(), kernel.data, kb);
    else memcpy(k.data(), kernel.data, kb);
  } else {
    kw = kh = 1; /* empty loop in the reference: sum = 0 */
    k[0] = 0;
  }
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, nb) : dst.data;
  gsh_filter_batch(d, s, src.w, src.h, 1, k.data(), kw, kh, norm);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, nb, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

void gs_downsample(struct gs_image dst, struct gs_image src) { /* ref :189 */
  GS_ASSERT(GS_VALID(src) && GS_VALID(dst) && dst.w == src.w / 2 && dst.h == src.h / 2);
  const size_t nb = (size_t)src.w * src.h, db = (size_t)dst.w * dst.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, db) : dst.data;
  gsh_downsample_batch(d, s, src.w, src.h, 1);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, db, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}

/* ---- geometry + template matching ------------------------------------------ */
void gs_crop(struct gs_image dst, struct gs_image src, struct gs_rect roi) { /* ref :154 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src) && roi.x + roi.w <= src.w && roi.y + roi.h <= src.h &&
            dst.w == roi.w && dst.h == roi.h);
  const size_t nb = (size_t)src.w * src.h, db = (size_t)dst.w * dst.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, db) : dst.data;
  GS_LAUNCH(k_crop, dim3((roi.w + 63) / 64, (roi.h + 3) / 4), dim3(64, 4), 0, ctx().s(), d, dst.w, dst.h, s,
            src.w, src.h, roi.x, roi.y, roi.w, roi.h);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, db, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}
void gs_copy(struct gs_image dst, struct gs_image src) { /* ref :160 */
  const struct gs_rect all = {0, 0, src.w, src.h};
  gs_crop(dst, src, all);
}
static void resize_common(struct gs_image dst, struct gs_image src, bool nearest) {
  const size_t nb = (size_t)src.w * src.h, db = (size_t)dst.w * dst.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, nb, SL_IN);
  const bool dhost = !is_dev(dst.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, db) : dst.data;
  const dim3 g((dst.w + 63) / 64, (dst.h + 3) / 4);
  if (nearest) GS_LAUNCH(k_resize_nn, g, dim3(64, 4), 0, ctx().s(), d, dst.w, dst.h, s, src.w, src.h);
  else GS_LAUNCH(k_resize, g, dim3(64, 4), 0, ctx().s(), d, dst.w, dst.h, s, src.w, src.h);
  if (dhost) GS_HIP(hipMemcpyAsync(dst.data, d, db, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}
void gs_resize_nn(struct gs_image dst, struct gs_image src) { /* ref :164 (asserts nothing there) */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src));
  resize_common(dst, src, true);
}
void gs_resize(struct gs_image dst, struct gs_image src) { /* ref :171 */
  GS_ASSERT(GS_VALID(dst) && GS_VALID(src));
  resize_common(dst, src, false);
}
void gs_match_template(struct gs_image img, struct gs_image tmpl, struct gs_image result) { /* ref :705 */
  GS_ASSERT(GS_VALID(img) && GS_VALID(tmpl) && GS_VALID(result));
  GS_ASSERT(img.w >= tmpl.w && img.h >= tmpl.h);
  GS_ASSERT(result.w == img.w - tmpl.w + 1 && result.h == img.h - tmpl.h + 1);
  const size_t ib = (size_t)img.w * img.h, tb = (size_t)tmpl.w * tmpl.h, rb = (size_t)result.w * result.h;
  const uint8_t *s = (const uint8_t *)stage_in(img.data, ib, SL_IN);
  const uint8_t *t = (const uint8_t *)stage_in(tmpl.data, tb, SL_AUX);
  const bool dhost = !is_dev(result.data);
  uint8_t *d = dhost ? (uint8_t *)ctx().scratch(SL_OUT, rb) : result.data;
  const dim3 g((result.w + 63) / 64, (result.h + 3) / 4);
  /* templates of 512 .. 32768 taps, 16 .. 257 wide, whose block fits the LDS: the cross term on the matrix cores (k_tmatch.h;
   * smaller ones are as fast on the dot-product kernels: 1280x720, 16 x 16: 39 vs 47 us).
   * gsh_tune key 20 = 1: the VALU dot-product kernels below; 2 / 3: always 64 x 128 tiles / always 32 x 64 tiles with split rows. */
  /* few 64 x 128 tiles (video-sized images): 32 x 64 tiles, the template rows split over the block's four waves */
  const bool tm_split = g_tune[20] == 3 || (g_tune[20] != 2 && (unsigned long long)((result.w + 127) / 128) * ((result.h + 63) / 64) < 512);
  const unsigned nkc = (tmpl.w + 31 + 31) / 32, istride = (tm_split ? 32 : 96) + 32 * nkc + 16, tstride = 32 * nkc + 48;
  const size_t tm_lds = std::max<size_t>((size_t)((tm_split ? 31 : 63) + tmpl.h) * istride + (size_t)tmpl.h * tstride + 16,
                                         tm_split ? 32768 : 0);
  if (g_tune[20] != 1 && tmpl.w >= 16 && nkc <= 9 && tmpl.h >= 4 && (tb >= 512 || g_tune[20] >= 2) && tb <= 32768 && tm_lds <= 150 * 1024 &&
      ib < 0x7fffffffull) {
    hipStream_t st = ctx().s();
    unsigned *rowp = (unsigned *)ctx().scratch(SL_II, (size_t)img.h * (img.w + 1) * 4);
    unsigned *s2 = (unsigned *)ctx().scratch(SL_PAD, rb * 4);
    uint8_t *tpad = (uint8_t *)ctx().scratch(SL_PRE, (size_t)tmpl.h * tstride + 16);
    unsigned *tsqp = (unsigned *)(tpad + (((size_t)tmpl.h * tstride + 3) & ~(size_t)3)); /* tstride is a multiple of 16 */
    GS_LAUNCH(k_tm_prep, dim3(1), dim3(1024), 0, st, t, tmpl.w, tmpl.h, tstride, tpad, tsqp);
    GS_LAUNCH(k_tm_rowprefix, dim3((img.h + 3) / 4), dim3(64, 4), 0, st, s, img.w, img.h, rowp);
    GS_LAUNCH(k_tm_colsq, dim3((result.w + 63) / 64, (result.h + kTmRun - 1) / kTmRun), dim3(64), 0, st, (const unsigned *)rowp,
              img.w, tmpl.w, tmpl.h, result.w, result.h, s2);
#ifndef GS_EMU
    /* more than the default 64 KB of dynamic LDS: a per-DEVICE attribute of the function (a thread that moved to
     * another device with gsh_set_device kept a thread-local "done" flag and the launch failed there) */
    static std::atomic<unsigned long long> lds_raised{0};
    const unsigned long long dev_bit = 1ull << ((unsigned)ctx().device & 63u);
    if (ctx().device >= 64 || !(lds_raised.load(std::memory_order_acquire) & dev_bit)) {
      GS_HIP(hipFuncSetAttribute((const void *)k_match_template_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      GS_HIP(hipFuncSetAttribute((const void *)k_match_template_mfma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      lds_raised.fetch_or(dev_bit, std::memory_order_release);
    }
#endif
    TmArgs ta{s, img.w, img.h, (const uint8_t *)tpad, (const unsigned *)tsqp, tmpl.w, tmpl.h, s2, d, result.w, result.h, nkc, istride, tstride};
    if (tm_split)
      GS_LAUNCH(k_match_template_mfma<4>, dim3((result.w + 63) / 64, (result.h + 31) / 32), dim3(256), tm_lds, st, ta);
    else
      GS_LAUNCH(k_match_template_mfma<1>, dim3((result.w + 127) / 128, (result.h + 63) / 64), dim3(256), tm_lds, st, ta);
  } else if (tmpl.w <= kTmplTile - 3) {
    unsigned long long *tsq = (unsigned long long *)ctx().scratch(SL_PFX, 8);
    GS_LAUNCH(k_sum_squares, dim3(1), dim3(256), 0, ctx().s(), t, (unsigned long long)tb, tsq);
    const size_t twp = ((size_t)tmpl.w + 3) & ~(size_t)3;
    const size_t lds = std::min<size_t>(twp * tmpl.h, (kTmplTile / twp) * twp);
    /* four results per thread when image rows start 4-byte aligned (see k_match_template4) */
    if (img.w % 4 == 0 && ((uintptr_t)s & 3) == 0 && (size_t)img.w * img.h < 0x7fffffffull)
      GS_LAUNCH(k_match_template4, dim3((result.w + 255) / 256, (result.h + 3) / 4), dim3(64, 4), lds, ctx().s(),
                s, img.w, img.h, t, tmpl.w, tmpl.h, (const unsigned long long *)tsq, d, result.w, result.h);
    else
      GS_LAUNCH(k_match_template, g, dim3(64, 4), lds, ctx().s(), s, img.w, img.h, t, tmpl.w, tmpl.h,
                (const unsigned long long *)tsq, d, result.w, result.h);
  } else {
    GS_LAUNCH(k_match_template_px, g, dim3(64, 4), 0, ctx().s(), s, img.w, img.h, t, tmpl.w, tmpl.h, d,
              result.w, result.h);
  }
  if (dhost) GS_HIP(hipMemcpyAsync(result.data, d, rb, hipMemcpyDeviceToHost, ctx().s()));
  finish(dhost);
}
struct gs_point gs_find_best_match(struct gs_image result) { /* ref :726 */
  GS_ASSERT(GS_VALID(result));
  const unsigned long long n = (unsigned long long)result.w * result.h;
  const uint8_t *s = (const uint8_t *)stage_in(result.data, (size_t)n, SL_IN);
  const unsigned blocks = (unsigned)((n + 2047) / 2048);
  unsigned long long *part = (unsigned long long *)ctx().scratch(SL_PFX, (size_t)blocks * 8);
  GS_LAUNCH(k_argmax_first, dim3(blocks), dim3(256), 0, ctx().s(), s, n, part);
  std::vector<unsigned long long> hp(blocks);
  GS_HIP(hipMemcpyAsync(hp.data(), part, (size_t)blocks * 8, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  unsigned long long best = 0;
  for (unsigned long long k : hp) best = std::max(best, k);
  struct gs_point p = {0, 0};
  if (best >> 32) { /* a zero maximum leaves the reference's initial {0,0} */
    const unsigned idx = 0xffffffffu - (unsigned)(best & 0xffffffffu);
    p.x = idx % result.w, p.y = idx / result.w;
  }
  return p;
}

void gs_histogram(struct gs_image img, unsigned hist[256]) { /* ref :199 */
  GS_ASSERT(GS_VALID(img) && hist != NULL);
  const size_t nb = (size_t)(img.w * img.h);
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  const bool hhost = !is_dev(hist);
  unsigned *dh = hhost ? (unsigned *)ctx().scratch(SL_HIST, 1024) : hist;
  launch_histogram(s, nb, 1, dh);
  if (hhost) GS_HIP(hipMemcpyAsync(hist, dh, 1024, hipMemcpyDeviceToHost, ctx().s()));
  finish(hhost);
}

uint8_t gs_otsu_threshold(struct gs_image img) { /* ref :205 */
  GS_ASSERT(GS_VALID(img));
  const size_t nb = (size_t)(img.w * img.h);
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  unsigned *dh = (unsigned *)ctx().scratch(SL_HIST, 1024);
  uint8_t *dt = (uint8_t *)ctx().scratch(SL_THR, 16);
  launch_otsu(s, img.w, img.h, 1, dh, dt);
  uint8_t t = 0;
  GS_HIP(hipMemcpyAsync(&t, dt, 1, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  return t;
}

void gs_threshold(struct gs_image img, uint8_t thresh) { /* ref :225 */
  GS_ASSERT(GS_VALID(img));
  const size_t nb = (size_t)(img.w * img.h);
  const bool host = !is_dev(img.data);
  uint8_t *d = host ? (uint8_t *)ctx().scratch(SL_IN, nb) : img.data;
  if (host) GS_HIP(hipMemcpyAsync(d, img.data, nb, hipMemcpyHostToDevice, ctx().s()));
  launch_threshold(d, nb, 1, nullptr, thresh);
  if (host) GS_HIP(hipMemcpyAsync(img.data, d, nb, hipMemcpyDeviceToHost, ctx().s()));
  finish(host);
}

void gs_integral(struct gs_image src, unsigned *ii) { /* ref :744 */
  GS_ASSERT(GS_VALID(src) && ii);
  const size_t np = (size_t)src.w * src.h;
  const uint8_t *s = (const uint8_t *)stage_in(src.data, np, SL_IN);
  const bool host = !is_dev(ii);
  unsigned *d = host ? (unsigned *)ctx().scratch(SL_II, np * 4) : ii;
  launch_integral(s, src.w, src.h, 1, d);
  if (host) GS_HIP(hipMemcpyAsync(ii, d, np * 4, hipMemcpyDeviceToHost, ctx().s()));
  finish(host);
}

/* The reference re-reads the caller's tables on every call (ref :790-835), so a cascade edited in
 * place, or freed and rebuilt at the same addresses, must take effect.  The flattened device copy
 * is therefore cached per calling thread keyed by a hash of the table CONTENTS (FNV-1a 64 over
 * ~7 KB: microseconds next to the launch), never by the struct's pointer values. */
static uint64_t cascade_content_hash(const struct gs_lbp_cascade *c) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  };
  const uint16_t dims[5] = {c->window_w, c->window_h, c->nfeatures, c->nweaks, c->nstages};
  mix(dims, sizeof dims);
  mix(c->features, (size_t)c->nfeatures * 4);
  mix(c->weak_feature_idx, (size_t)c->nweaks * 2);
  mix(c->weak_left_val, (size_t)c->nweaks * 4);
  mix(c->weak_right_val, (size_t)c->nweaks * 4);
  mix(c->weak_subset_offset, (size_t)c->nweaks * 2);
  mix(c->weak_num_subsets, (size_t)c->nweaks * 2);
  unsigned nsub = 0;
  for (unsigned i = 0; i < c->nweaks; i++)
    nsub = std::max(nsub, (unsigned)c->weak_subset_offset[i] + c->weak_num_subsets[i]);
  mix(c->subsets, (size_t)nsub * 4);
  mix(c->stage_weak_start, (size_t)c->nstages * 2);
  mix(c->stage_nweaks, (size_t)c->nstages * 2);
  mix(c->stage_threshold, (size_t)c->nstages * 4);
  return h;
}
static gsh_cascade *cached_cascade(const struct gs_lbp_cascade *c) {
  Ctx &cx = ctx();
  const uint64_t h = cascade_content_hash(c);
  if (cx.dropin_cascade && cx.dropin_cascade_hash == h) return cx.dropin_cascade;
  if (cx.dropin_cascade) gsh_cascade_destroy(cx.dropin_cascade);
  cx.dropin_cascade = gsh_cascade_create(c);
  cx.dropin_cascade_hash = h;
  return cx.dropin_cascade;
}

unsigned gs_lbp_detect(const struct gs_lbp_cascade *c, const unsigned *ii, unsigned iw,
                       unsigned ih, struct gs_rect *rects, unsigned max_rects, float scale_factor,
                       float min_scale, float max_scale, int step) { /* ref :815 */
  GS_ASSERT(c && ii && iw > 0 && ih > 0 && (rects || max_rects == 0));
  if (max_rects == 0) return 0;
  gsh_cascade *dc = cached_cascade(c);
  const size_t np = (size_t)iw * ih;
  const unsigned *dii = (const unsigned *)stage_in(ii, np * 4, SL_II);
  const bool rhost = !is_dev(rects);
  unsigned *dr = rhost ? (unsigned *)ctx().scratch(SL_OUT, (size_t)max_rects * 16) : (unsigned *)rects;
  unsigned *dcnt = (unsigned *)ctx().scratch(SL_TOT, 16);
  launch_lbp_unpadded(dc, dii, iw, ih, 1, dr, dcnt, max_rects, scale_factor, min_scale, max_scale,
                      step);
  unsigned n = 0;
  GS_HIP(hipMemcpyAsync(&n, dcnt, 4, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  if (rhost && n) GS_HIP(hipMemcpy(rects, dr, (size_t)n * 16, hipMemcpyDeviceToHost));
  return n;
}

unsigned gs_lbp_window(const struct gs_lbp_cascade *c, const unsigned *ii, unsigned iw, unsigned ih,
                       int x, int y, float scale) { /* ref :790 */
  GS_ASSERT(c && ii);
  const int win_w = (int)((int)c->window_w * scale), win_h = (int)((int)c->window_h * scale);
  if (x + win_w > (int)iw || y + win_h > (int)ih) return 0; /* ref :793 */
  if (x < 0 || y < 0 || win_w <= 0 || win_h <= 0) return 0;
  gsh_cascade *dc = cached_cascade(c);
  hipStream_t st = ctx().s();
  /* (win_w+1) x (win_h+1) zero-bordered sub-table around the window: the cascade only ever
   * forms D + A - B - C differences, so absolute table values carry over unchanged */
  const unsigned S = (unsigned)win_w + 1, R = (unsigned)win_h + 1;
  unsigned *tab = (unsigned *)ctx().scratch(SL_PAD, (size_t)S * R * 4 + 64);
  GS_HIP(hipMemsetAsync(tab, 0, (size_t)S * R * 4, st));
  const unsigned cx = x > 0 ? 1 : 0, cy = y > 0 ? 1 : 0;
  const unsigned *src0 = ii + (size_t)(y - (int)cy) * iw + (x - (int)cx);
  GS_HIP(hipMemcpy2DAsync(tab + (size_t)(1 - cy) * S + (1 - cx), (size_t)S * 4, src0, (size_t)iw * 4,
                          (size_t)(win_w + cx) * 4, (size_t)(win_h + cy),
                          is_dev(ii) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  std::vector<LbpGeom> geom(dc->nweaks);
  for (unsigned wi = 0; wi < dc->nweaks; wi++) {
    const int fi = dc->weak_feature_idx[wi];
    int fx = (int)((int)dc->features[fi * 4 + 0] * scale), fy = (int)((int)dc->features[fi * 4 + 1] * scale);
    int fw = (int)((int)dc->features[fi * 4 + 2] * scale), fh = (int)((int)dc->features[fi * 4 + 3] * scale);
    if (fw < 1) fw = 1;
    if (fh < 1) fh = 1;
    geom[wi] = LbpGeom{(fy * (int)S + fx) * 4, fw * 4, fh * (int)S * 4, 0};
  }
  LbpGeom *dg = (LbpGeom *)ctx().scratch(SL_TAB, std::max<size_t>(1, geom.size()) * sizeof(LbpGeom));
  GS_HIP(hipMemcpyAsync(dg, geom.data(), geom.size() * sizeof(LbpGeom), hipMemcpyHostToDevice, st));
  unsigned *dout = (unsigned *)ctx().scratch(SL_TOT, 16);
  LbpArgs a;
  memset(&a, 0, sizeof a);
  a.padded = tab, a.frame_stride = (size_t)S * R, a.S = S, a.limit_bytes = (unsigned)(((size_t)S * R - 1) * 4), a.step = 1;
  a.nweaks = dc->nweaks, a.nstages = dc->nstages, a.nsub = dc->nsub;
  a.weak = dc->d_weak, a.stage = dc->d_stage, a.subsets = dc->d_subsets;
  const size_t lds = (size_t)dc->nstages * sizeof(LbpStage) +
                     (size_t)dc->nweaks * (sizeof(LbpWeak) + sizeof(LbpGeom)) + (size_t)dc->nsub * 4;
  GS_LAUNCH(k_lbp_single, dim3(1), dim3(64), lds, st, a, (const LbpGeom *)dg, dout);
  unsigned r = 0;
  GS_HIP(hipMemcpyAsync(&r, dout, 4, hipMemcpyDeviceToHost, st));
  ctx().sync();
  return r;
}

unsigned gs_fast(struct gs_image img, struct gs_image scoremap, struct gs_keypoint *kps,
                 unsigned nkps, unsigned threshold) { /* ref :482 */
  GS_ASSERT(GS_VALID(img) && kps && nkps > 0);
  const unsigned w = img.w, h = img.h;
  if (w < 7 || h < 7) return 0;
  /* The reference writes the map through gs_set and reads it through gs_get (ref :512, :518-524), so a
   * map of another size -- or no map at all -- is legal: positions outside it are never written and
   * read 0.  Same here: the kernels run on an image-sized device map M that starts as the caller's
   * map where the two overlap (0 elsewhere); positions outside the caller's map are zeroed again
   * between the two passes; the overlap is copied back. */
  if (!GS_VALID(scoremap)) return 0; /* every gs_get(scoremap) is 0: the NMS pass skips every pixel */
  const size_t nb = (size_t)w * h;
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  const bool same = scoremap.w == w && scoremap.h == h;
  const bool mhost = !is_dev(scoremap.data);
  const unsigned ow = std::min(w, scoremap.w), oh = std::min(h, scoremap.h); /* overlap */
  uint8_t *dm = (mhost || !same) ? (uint8_t *)ctx().scratch(SL_AUX, nb) : scoremap.data;
  /* NMS reads the caller's 3-px frame (ref :524): ship the whole map in */
  if (same) {
    if (mhost) GS_HIP(hipMemcpyAsync(dm, scoremap.data, nb, hipMemcpyHostToDevice, ctx().s()));
  } else {
    GS_HIP(hipMemsetAsync(dm, 0, nb, ctx().s()));
    GS_HIP(hipMemcpy2DAsync(dm, w, scoremap.data, scoremap.w, ow, oh,
                            mhost ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx().s()));
  }
  const bool khost = !is_dev(kps);
  unsigned *dk = khost ? (unsigned *)ctx().scratch(SL_KPS, (size_t)nkps * 48) : (unsigned *)kps;
  unsigned *dcnt = (unsigned *)ctx().scratch(SL_TOT, 16);
  launch_fast(s, dm, w, h, 1, dk, dcnt, nkps, threshold, same ? 0u : scoremap.w, same ? 0u : scoremap.h);
  unsigned n = 0;
  GS_HIP(hipMemcpyAsync(&n, dcnt, 4, hipMemcpyDeviceToHost, ctx().s()));
  if (same) {
    if (mhost) GS_HIP(hipMemcpyAsync(scoremap.data, dm, nb, hipMemcpyDeviceToHost, ctx().s()));
  } else {
    GS_HIP(hipMemcpy2DAsync(scoremap.data, scoremap.w, dm, w, ow, oh,
                            mhost ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx().s()));
  }
  ctx().sync();
  if (khost && n) GS_HIP(hipMemcpy(kps, dk, (size_t)n * 48, hipMemcpyDeviceToHost));
  return n;
}

/* image patch [x-r, x+r] x [y-r, y+r] zero-filled outside the image, on the device */
static const uint8_t *stage_patch(struct gs_image img, int x, int y, int r, int slot) {
  const int side = 2 * r + 1;
  uint8_t *d = (uint8_t *)ctx().scratch(slot, (size_t)side * side);
  hipStream_t st = ctx().s();
  GS_HIP(hipMemsetAsync(d, 0, (size_t)side * side, st));
  const int xa = std::max(0, x - r), xb = std::min((int)img.w - 1, x + r);
  const int ya = std::max(0, y - r), yb = std::min((int)img.h - 1, y + r);
  if (xa <= xb && ya <= yb)
    GS_HIP(hipMemcpy2DAsync(d + (size_t)(ya - (y - r)) * side + (xa - (x - r)), side,
                            img.data + (size_t)ya * img.w + xa, img.w, (size_t)(xb - xa + 1),
                            (size_t)(yb - ya + 1),
                            is_dev(img.data) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  return d;
}

float gs_compute_orientation(struct gs_image img, unsigned x, unsigned y, unsigned r) { /* ref :608 */
  GS_ASSERT(GS_VALID(img) && x >= r && y >= r && x < img.w - r && y < img.h - r);
  hipStream_t st = ctx().s();
  const uint8_t *patch = stage_patch(img, (int)x, (int)y, (int)r, SL_AUX2);
  if (r > kOrientExactR) { /* partial sums can pass 2^24: the reference's float32 order, one thread */
    float *df = (float *)ctx().scratch(SL_MOM, 16);
    GS_LAUNCH(k_orient_moments_seq, dim3(1), dim3(1), 0, st, patch, 2 * r + 1, 2 * r + 1, r, r, r, df);
    float mf[2];
    GS_HIP(hipMemcpyAsync(mf, df, 8, hipMemcpyDeviceToHost, st));
    ctx().sync();
    return atan2f(mf[0], mf[1]);
  }
  unsigned pt[2] = {r, r};
  unsigned *dp = (unsigned *)ctx().scratch(SL_KIN, 16);
  int *dm = (int *)ctx().scratch(SL_MOM, 16);
  GS_HIP(hipMemcpyAsync(dp, pt, 8, hipMemcpyHostToDevice, st));
  GS_LAUNCH(k_orient_moments, dim3(1), dim3(64), 0, st, patch, 2 * r + 1, 2 * r + 1,
            (const unsigned *)dp, 2u, r, dm, (const unsigned *)nullptr);
  int m[2];
  GS_HIP(hipMemcpyAsync(m, dm, 8, hipMemcpyDeviceToHost, st));
  ctx().sync();
  return atan2f((float)m[0], (float)m[1]); /* ref :620 -> libm, as the reference (ref :100) */
}

void gs_brief_descriptor(struct gs_image img, struct gs_keypoint *kp) { /* ref :623 */
  GS_ASSERT(GS_VALID(img) && kp);
  hipStream_t st = ctx().s();
  const int R = 22; /* |pattern| <= 15 rotated reaches <= 21 px */
  gs_keypoint k;
  if (is_dev(kp)) gsh_download(&k, kp, sizeof k);
  else k = *kp;
  const uint8_t *patch = stage_patch(img, (int)k.pt.x, (int)k.pt.y, R, SL_AUX2);
  const float angle = k.angle;
  KpIn in{(unsigned)R, (unsigned)R, sinf(angle), sinf((float)(angle + 1.57079f))};
  KpIn *dk = (KpIn *)ctx().scratch(SL_KIN, sizeof in);
  uint32_t *dd = (uint32_t *)ctx().scratch(SL_DESC, 32);
  GS_HIP(hipMemcpyAsync(dk, &in, sizeof in, hipMemcpyHostToDevice, st));
  GS_LAUNCH(k_brief, dim3(1), dim3(256), 0, st, patch, 2u * R + 1, 2u * R + 1, (const KpIn *)dk, dd);
  GS_HIP(hipMemcpyAsync(k.descriptor, dd, 32, hipMemcpyDeviceToHost, st));
  ctx().sync();
  if (is_dev(kp)) gsh_upload(kp, &k, sizeof k);
  else memcpy(kp->descriptor, k.descriptor, 32);
}

unsigned gs_orb_extract(struct gs_image img, struct gs_keypoint *kps, unsigned nkps,
                        unsigned threshold, uint8_t *scoremap_buffer) { /* ref :651 */
  GS_ASSERT(GS_VALID(img) && kps && nkps > 0 && scoremap_buffer);
  const unsigned w = img.w, h = img.h;
  if (w < 7 || h < 7) return 0;
  const size_t nb = (size_t)w * h;
  const uint8_t *s = (const uint8_t *)stage_in(img.data, nb, SL_IN);
  const bool mhost = !is_dev(scoremap_buffer);
  uint8_t *dm = mhost ? (uint8_t *)ctx().scratch(SL_AUX, nb) : scoremap_buffer;
  if (mhost) GS_HIP(hipMemcpyAsync(dm, scoremap_buffer, nb, hipMemcpyHostToDevice, ctx().s()));
  const bool khost = !is_dev(kps);
  std::vector<gs_keypoint> tmp;
  gs_keypoint *out = kps;
  if (!khost) {
    tmp.resize(nkps);
    out = tmp.data();
  }
  OrbLevel L{s, w, h, dm, out, nkps, 0};
  orb_extract_levels(&L, 1, threshold);
  const unsigned n = L.got;
  if (mhost) GS_HIP(hipMemcpyAsync(scoremap_buffer, dm, nb, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  if (!khost && n) gsh_upload(kps, out, (size_t)n * sizeof(gs_keypoint));
  return n;
}

unsigned gs_match_orb(const struct gs_keypoint *kps1, unsigned n1, const struct gs_keypoint *kps2,
                      unsigned n2, struct gs_match *matches, unsigned max_matches,
                      float max_distance) { /* ref :680 */
  GS_ASSERT(kps1 && kps2 && matches);
  if (n1 == 0 || max_matches == 0) return 0;
  const uint32_t *d1 = (const uint32_t *)stage_in(kps1, (size_t)n1 * 48, SL_IN);
  const uint32_t *d2 = n2 ? (const uint32_t *)stage_in(kps2, (size_t)n2 * 48, SL_AUX)
                          : (const uint32_t *)ctx().scratch(SL_AUX, 48);
  const bool mhost = !is_dev(matches);
  unsigned *dm = mhost ? (unsigned *)ctx().scratch(SL_OUT, (size_t)max_matches * 12) : (unsigned *)matches;
  unsigned *dcnt = (unsigned *)ctx().scratch(SL_TOT, 16);
  launch_match(d1, n1, d2, n2, dm, dcnt, max_matches, max_distance);
  unsigned n = 0;
  GS_HIP(hipMemcpyAsync(&n, dcnt, 4, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  if (mhost && n) GS_HIP(hipMemcpy(matches, dm, (size_t)n * 12, hipMemcpyDeviceToHost));
  return n;
}

}  /* extern "C" */
