/*
 * include/grayskull_hip.h -- device-resident / batched entry points of
 * libgrayskull_hip.so (MI355X, gfx950).  Plain C ABI: pointers and sizes only.
 *
 * The reference API (grayskull.h) is one image per call on caller-owned host
 * memory.  A 4K frame is ~2 us of HBM traffic, the same as one kernel boundary, so
 * throughput needs (a) images that already live in HBM and (b) many frames per
 * launch.  The gsh_* functions are that: every pointer is a DEVICE pointer, a
 * batch is `n` frames of w*h bytes laid out back to back, calls are enqueued on
 * the calling thread's stream and return immediately (no host sync) unless noted.
 *
 * Each batch function computes, per frame, exactly what the cited reference
 * function computes (bit-exact; tests/ compare against the oracle).
 */
#ifndef GRAYSKULL_HIP_H
#define GRAYSKULL_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "grayskull.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ------------------------------------------------------------------- */
const char *gsh_version(void);
int gsh_device_count(void);               /* 0 when no GPU is visible                 */
void gsh_set_device(int ordinal);         /* per calling thread; default device 0      */
void gsh_set_stream(void *hip_stream);    /* NULL => the library's own per-thread one   */
void *gsh_get_stream(void);
void gsh_set_async(int on);               /* drop-in gs_* calls on device pointers skip
                                             the final stream sync when on             */
void gsh_sync(void);                      /* hipStreamSynchronize(current stream)       */
/* ---- launch tuning.  gsh_tune(key, value) sets one entry of a PROCESS-WIDE table of launch heuristics (every entry an
 * atomic; nothing in a product path writes them): measurement scripts and the test-suite use it to force the paths a
 * heuristic would not take on a given input.  RESULTS NEVER DEPEND ON ANY KEY.  (The two probes that did change results --
 * the cascade without rect emission, gs_sobel without its column reads -- exist only in builds with -DGS_EXPERIMENT, see
 * below.)  0 is every key's default. */
enum gsh_tune_key {
  GSH_TUNE_STRIP_BAND_ROWS = 0,   /* rows per band of the strip kernels (0 = auto) */
  GSH_TUNE_STRIP_BLOCK_SHAPE = 1, /* 0: 64x4, 1: 256x1, 2: 128x2, 3: by frame width (default) */
  GSH_TUNE_2 = 2,                 /* reserved (default 1) */
  GSH_TUNE_NO_FUSED_PIPELINE = 3, /* 1: gsh_edge_pipeline_batch on the separate per-call kernels */
  GSH_TUNE_LBP_PHASE_PRESET = 4,  /* k_lbp_cascade: preset of the stages at which a block re-packs survivors (1: never; >= 1000: custom split) */
  GSH_TUNE_PIPELINE_CHUNK = 5,    /* frames per chunk of gsh_edge_pipeline_batch's internal overlap (0 = 32, negative = never split) */
  GSH_TUNE_COMPARE = 6,           /* 1 generic two-pass gs_integral, 2 block-per-band gs_integral, 3 integral-image route for gs_blur(r > 3) /
                                     gs_adaptive_threshold, 4 the any-radius box kernel also for radii <= 16, 5 ... for ragged rows only, 6 / 7 k_box_edge always on the caller's / on the side stream,
                                     8 gs_integral without the streaming loads of batches beyond the Infinity Cache */
  GSH_TUNE_FAST_SCORE = 7,        /* 0: k_fast_score_q4 (LDS tile, candidates queued), 2: k_fast_score_px (one global byte load per ring pixel) */
  GSH_TUNE_FRAMES_PER_LAUNCH = 8, /* test hook for the batch splitting of every launcher */
  GSH_TUNE_LBP_ADAPTIVE = 9,      /* k_lbp_cascade: max stages + 16 * tenths [+ later points] of the first re-packing point */
  GSH_TUNE_HIST_TRIPS = 10,       /* trips per block gs_histogram aims at; -1: default-policy loads also for batches beyond the Infinity Cache (A/B) */
  GSH_TUNE_HIST_BLOCKS = 11,      /* its blocks per frame */
  GSH_TUNE_HIST_PIECE = 12,       /* bytes per histogram piece (test hook for images above 1 GiB) */
  GSH_TUNE_LBP_XCD = 13,          /* chunk / tile -> XCD mapping of the LBP kernels: 1 dispatch order, 2 XCD-aware always (0: k_lbp_cascade by table size, k_lbp_tile in dispatch order) */
  GSH_TUNE_LBP_KERNEL = 14,       /* 0: per scale by rule (k_lbp_tile with the tile shape the scale's LDS footprint allows, else
                                     k_lbp_cascade), 1: k_lbp_cascade for every scale, -1: the rule without its one-block-per-CU
                                     fallback, 2 + i: tile shape i of k_lbp_tile wherever it fits */
  GSH_TUNE_LBP_TILE_SWITCH = 15,  /* k_lbp_tile: first + 16 * tenths -- dense stages [0, first), then while more than tenths/10 of a
                                     wave's windows live, then one lane per (window, classifier) pair */
  GSH_TUNE_EXPERIMENT_16 = 16,    /* GS_EXPERIMENT builds only: gs_lbp_detect runs its kernels but emits nothing */
  GSH_TUNE_LBP_ONE_LANE = 17,     /* 1: k_lbp_cascade evaluates re-packed windows one per lane instead of one per quad */
  GSH_TUNE_STRIP_XCD = 18,        /* band -> XCD mapping of the strip kernels / tile -> XCD mapping of the gs_fast score pass: 1 dispatch
                                     order, 2 XCD-aware always */
  GSH_TUNE_FAST_NMS = 19,         /* 0: k_fast_nms_sparse behind the score kernel's bitmap, 1: k_fast_nms item by item */
  GSH_TUNE_TMATCH = 20,           /* 1: gs_match_template on the VALU dot-product kernels; 2 / 3: matrix-core kernel with 64 x 128 / 32 x 64
                                     tiles whatever the image size; 4 / 5: window sums of squares always by passes / always from the table */
  GSH_TUNE_STRIP_ROUND3_RULE = 21,/* 1: strip kernels only for whole 16-px strips at 16-byte aligned addresses */
  GSH_TUNE_EXPERIMENT_22 = 22,    /* GS_EXPERIMENT builds only: gs_sobel without the reads that preserve columns 0 / w-1 */
  GSH_TUNE_EXPERIMENT_23 = 23,    /* GS_EXPERIMENT builds only: the strip-copy probe keeps the stencils' halo load */
  GSH_TUNE_STRIP_REALIGN = 24     /* realigning strip flavour: 1 never, 2 always, 0 by address phase and width */
};
void gsh_tune(int key, int value);
/* measurement aid for bench.py: while on, gsh_edge_pipeline_batch brackets every launch of its
 * fused blur+sobel+histogram kernel with HIP events on the stream it is launched on (up to 4096
 * launches between reads; on > 1 pre-creates that many event pairs so none is created inside a
 * timed region); gsh_profile_read synchronises, returns how many launches were bracketed since
 * the last read and their summed duration in milliseconds. */
void gsh_profile(int on);
unsigned gsh_profile_read(double *total_ms);
#ifdef GS_EXPERIMENT
/* Experiment builds only (make experiment -> build_variants/libgs_experiment.so; the release library does not export it):
 * strip-kernel traffic pattern with no arithmetic (access-pattern ceiling).  The same builds honour the result-changing
 * keys 16 / 22 / 23 and the compile-time hooks GS_FUSED_SPARE, GS_FUSED_VGPR_ATTR, GS_EVENT_FLAGS, GS_ORDER_EVENT_FLAGS,
 * GS_LOAD_AUX, GS_STORE_AUX, GS_LBP_PREFETCH, GS_MAD2_OPAQUE, GS_LBP_TILE_ODD_STRIDE. */
void gsh_probe_strip_copy(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n);
#endif
void gsh_shutdown(void);                  /* free this thread's scratch + stream        */

void *gsh_malloc(size_t bytes);           /* hipMalloc; aborts on failure               */
void gsh_free(void *dev);
/* page-locked host memory (hipHostMalloc): gsh_upload / gsh_download and host-pointer gs_* calls
 * on it move at the full PCIe rate without the driver's pageable staging copy; aborts on failure */
void *gsh_host_alloc(size_t bytes);
void gsh_host_free(void *host);
void gsh_memset(void *dev, int byte, size_t bytes);             /* stream-ordered      */
void gsh_upload(void *dev, const void *host, size_t bytes);     /* synchronous         */
void gsh_download(void *host, const void *dev, size_t bytes);   /* synchronous         */
int gsh_is_device_ptr(const void *p);

/* ---- stencils over a batch (ref grayskull.h:268, :306, :303-304) ----------------- */
void gsh_blur_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                    unsigned radius);
void gsh_sobel_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n);
void gsh_erode_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n);
void gsh_dilate_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n);

/* ---- histogram / otsu / threshold (ref :199, :205, :225) ------------------------- */
/* hist: n x 256 u32 (device).  thr: n x u8 (device). */
void gsh_histogram_batch(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist);
void gsh_otsu_batch(const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned *hist_scratch,
                    uint8_t *thr);
void gsh_threshold_batch(uint8_t *img, unsigned w, unsigned h, unsigned n, uint8_t thresh);
void gsh_threshold_batch_dev(uint8_t *img, unsigned w, unsigned h, unsigned n, const uint8_t *thr);

/* gs_blur(src, radius) followed by gs_sobel into a zeroed dst, per frame, in one pass over the frame
 * for radius 1..3 (the blurred image stays in registers; ref :268, :306) */
void gsh_blur_sobel_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                          unsigned radius);

/* config-2 chain per frame: blur(radius) -> sobel (dst frame pre-zeroed) -> otsu -> threshold.
 * dst: n*w*h bytes, thr: the n Otsu thresholds.  tmp: n*w*h bytes that receive the blurred
 * frames, or NULL when the caller does not need them -- then blur, sobel and the histogram run
 * as ONE fused kernel and the blurred image never touches memory (same dst / thr bytes). */
void gsh_edge_pipeline_batch(uint8_t *dst, uint8_t *tmp, const uint8_t *src, unsigned w, unsigned h,
                             unsigned n, unsigned radius, unsigned *hist_scratch, uint8_t *thr);

/* ---- integral image + LBP cascade (ref :744, :815) ------------------------------- */
/* ii: n frames of w*h u32, same unpadded layout as the reference. */
void gsh_integral_batch(const uint8_t *src, unsigned w, unsigned h, unsigned n, unsigned *ii);

/* Cascade tables are flattened once into device memory. `c` points at HOST tables. */
typedef struct gsh_cascade gsh_cascade;
gsh_cascade *gsh_cascade_create(const struct gs_lbp_cascade *c);
void gsh_cascade_destroy(gsh_cascade *dc);

/* Per frame f: rects[f*max_rects ...] gets the first min(count,max_rects) detections in the
 * reference's (scale, y, x) order, counts[f] that number.  ii as produced by gsh_integral_batch.
 * rects/counts are device pointers.  Stream-ordered. */
void gsh_lbp_detect_batch(const gsh_cascade *dc, const unsigned *ii, unsigned iw, unsigned ih,
                          unsigned n, struct gs_rect *rects, unsigned *counts, unsigned max_rects,
                          float scale_factor, float min_scale, float max_scale, int step);
/* Measurement aid: while `counter_dev` (FOUR device u64, caller-zeroed) is set, every cascade launch
 * of this thread adds [0] the number of windows it really evaluated -- chunks skipped because
 * max_rects detections precede them in scan order (ref :819-823) are not counted --, [1] the
 * number of weak classifiers evaluated, summed over windows (every window pays the classifiers of the stages
 * it enters, like the reference), [2] the dword table loads the kernels issued, summed over lanes, [3] nothing
 * any more (round 3's stage prefilter counted its windows here; the buffer keeps four entries so that callers of
 * either round stay inside their allocation).  NULL = off (the default kernels). */
void gsh_lbp_count_evaluated(unsigned long long *counter_dev);
/* number of windows gs_lbp_detect visits for this geometry (for Mwin/s reporting) */
uint64_t gsh_lbp_window_count(const struct gs_lbp_cascade *c, unsigned iw, unsigned ih,
                              float scale_factor, float min_scale, float max_scale, int step);

/* ---- FAST / ORB / matching (ref :482, :651, :680) -------------------------------- */
/* pass 1 of gs_fast alone (ref :487-513): the FAST-9 score map of n frames (interior pixels; the 3-px frame is not written,
 * ref :489); w, h >= 7 */
void gsh_fast_score_batch(uint8_t *score, const uint8_t *img, unsigned w, unsigned h, unsigned n, unsigned threshold);
/* kps: n*nkps records (device), counts: n (device).  scoremap: n frames; only the interior is
 * written, the 3-px frame is read by the NMS exactly like the reference does. */
void gsh_fast_batch(const uint8_t *img, uint8_t *scoremap, unsigned w, unsigned h, unsigned n,
                    struct gs_keypoint *kps, unsigned *counts, unsigned nkps, unsigned threshold);

/* Device image -> HOST keypoints.  Synchronous (orientation uses the host libm, like the
 * reference: grayskull.h:100-101).  Returns the number of keypoints written. */
unsigned gsh_orb_extract(const uint8_t *img_dev, unsigned w, unsigned h, uint8_t *scoremap_dev,
                         struct gs_keypoint *kps_host, unsigned nkps, unsigned threshold);

/* gs_orb_extract for n frames of one size (frames w*h bytes apart): frame f's keypoints go to
 * kps_host[f*nkps ...], their number to counts_host[f].  scoremap_dev: n frames, same role as in
 * gsh_fast_batch.  Two host round trips for the whole batch.  Synchronous. */
void gsh_orb_extract_batch(const uint8_t *img_dev, unsigned w, unsigned h, unsigned n,
                           uint8_t *scoremap_dev, struct gs_keypoint *kps_host, unsigned *counts_host,
                           unsigned nkps, unsigned threshold);
/* gs_orb_extract for n device frames with NO host round trip: the selection (stable sort by response,
 * border filter, cap) and the trig run on the device.  The trig is the reference's GS_NO_STDLIB pair
 * (ref :70-88, the polynomials its wasm build uses), so results equal the reference header compiled
 * with -DGS_NO_STDLIB bit for bit -- angles and descriptors differ from the libm build's, like the
 * reference's own two builds differ.  kps_dev: n x nkps records, counts_dev: n; stream-ordered. */
void gsh_orb_extract_batch_nostdlib(const uint8_t *img_dev, unsigned w, unsigned h, unsigned n,
                                    uint8_t *scoremap_dev, struct gs_keypoint *kps_dev, unsigned *counts_dev,
                                    unsigned nkps, unsigned threshold);

/* The reference's ORB caller (examples/nanomagick/nanomagick.c:245-290, extract_pyramid_orb_nm) with
 * every pyramid level resident on the device: up to 4 levels, each gs_downsample (ref :189) of the
 * previous, stopping before a level narrower or lower than 32; nkps / n_levels keypoints per level
 * (the last level takes the remainder); coordinates scaled back by 2^level.  buffer_dev has the
 * reference's layout -- levels 1.. back to back, then one scoremap per level -- and its bytes are
 * the caller's (the NMS reads the never-written 3-px scoremap frames, ref :524).  Synchronous. */
size_t gsh_orb_pyramid_buffer_bytes(unsigned w, unsigned h, unsigned n_levels);
unsigned gsh_orb_extract_pyramid(const uint8_t *img_dev, unsigned w, unsigned h, uint8_t *buffer_dev,
                                 struct gs_keypoint *kps_host, unsigned nkps, unsigned threshold,
                                 unsigned n_levels);

/* All pointers device; matches: max_matches records; count: 1 u32.  Stream-ordered. */
void gsh_match_orb_dev(const struct gs_keypoint *k1, unsigned n1, const struct gs_keypoint *k2,
                       unsigned n2, struct gs_match *matches, unsigned *count,
                       unsigned max_matches, float max_distance);

/* ---- "next" rows (ref :230, :255, :189) ------------------------------------------ */
void gsh_adaptive_threshold_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                                  unsigned n, unsigned radius, int c);
void gsh_filter_batch(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned n,
                      const int8_t *kernel_host, unsigned kw, unsigned kh, unsigned norm);
void gsh_downsample_batch(uint8_t *dst, const uint8_t *src, unsigned sw, unsigned sh, unsigned n);

/* ---- synthetic frames + checksums on device (SURVEY.md 8c generator) ------------- */
/* frame f = synth(w, h, seed0 + f): bit-identical to the CPU generator. */
void gsh_synth_batch(uint8_t *dst, unsigned w, unsigned h, unsigned n, uint32_t seed0);
/* sums[f] = order-independent 64-bit checksum of frame f (sum of (i+1)*FNVmix(byte)) */
void gsh_checksum_batch(const uint8_t *img, size_t frame_bytes, unsigned n, uint64_t *sums);

/* ---- multi-GPU control plane for one-process host programs ---------
 * One host thread per device (gsh_set_device) and one communicator per device, created together by
 * ncclCommInitAll over the listed devices (devices == NULL: 0 .. ndev-1).  RCCL over xGMI carries control traffic
 * only -- the cascade blob, per-file counts and checksums, the closing max of the elapsed time; frames shard by
 * file / frame and never cross GPUs (the reference has no counterpart: grayskull.h holds no state across images).
 * librccl.so is dlopen()ed by gsh_comm_init_all: ndev == 1 takes local copies (GS_COMM_RCCL=1 asks for a one-rank RCCL
 * communicator), ndev > 1 without a usable librccl (or with GS_COMM_BACKEND=host) a host-rendezvous backend: the KB-scale
 * payloads bounce through host memory, the worker threads meet at a rendezvous.
 * Every collective takes DEVICE buffers, must be called by the thread that drives that communicator's device and is
 * enqueued on that thread's stream (gsh_sync() before the host reads a result).  Precondition failures and RCCL
 * errors abort like everything else here. */
typedef struct gsh_comm gsh_comm;
int gsh_comm_init_all(gsh_comm **comms, int ndev, const int *devices); /* 0 (a backend is always found) */
void gsh_comm_destroy_all(gsh_comm **comms, int ndev);
int gsh_comm_rank(const gsh_comm *c);
int gsh_comm_world(const gsh_comm *c);
const char *gsh_comm_backend(const gsh_comm *c); /* "rccl 2.x.y (ncclCommInitAll, N ranks)" | "local copies ..." */
void gsh_comm_broadcast(gsh_comm *c, void *buf_dev, size_t bytes, int root);
void gsh_comm_all_gather(gsh_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
void gsh_comm_all_reduce_u64(gsh_comm *c, unsigned long long *buf_dev, size_t n, int op); /* op: 0 sum, 1 max; in place */
void gsh_comm_all_reduce_f64(gsh_comm *c, double *buf_dev, size_t n, int op);

#ifdef __cplusplus
}
#endif
#endif /* GRAYSKULL_HIP_H */
