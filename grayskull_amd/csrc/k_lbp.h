/*
 * k_lbp.h -- gs_lbp_detect / gs_lbp_window (grayskull.h:769-835) on gfx950.
 *
 * One thread = one detection window; a block covers one chunk (2048 consecutive windows of
 * ONE scale, row-major like the reference's y/x loops), so every wave evaluates the same weak
 * classifier of the same scale at the same time: feature geometry, leaf values and stage
 * thresholds are wave-uniform (scalar loads), only the 16 integral-image corners and the
 * 256-bit subset lookup are per-lane.  The integral image is read in the zero-bordered
 * (w+1) x (h+1) layout (k_integral_pad), so each of the nine 3x3-cell sums is
 * D + A - B - C on unguarded loads and the 36 loads of the reference collapse to the 16
 * distinct corners.  Stage sums are sequential float32 adds in weak order (ref :796-810).
 *
 * Not HBM-bound: the 8.3 MB table sits in L2/Infinity Cache; the bound is gather rate and
 * divergence (63 % of windows die in stage 0).  Survivors go through k_compact.h so the output
 * order is the reference's (scale, y, x) with the max_rects cap.
 */
#ifndef GS_K_LBP_H
#define GS_K_LBP_H
#include "k_compact.h"
#include "lbp_types.h"

namespace gs {

struct LbpWeak { float left, right; unsigned sub_off, nsub; };
/* truth: 0, or 1 + the word offset of the stage's TRUTH TABLE in LbpArgs::truth -- bit P of it says whether the stage passes
 * when bit i of P is classifier i's subset-lookup result, i.e. !(sum < threshold) with the reference's sequential float32
 * adds of left / right in weak order (ref :796-810) carried out on the host for all 2^count patterns (stages of <= 12
 * classifiers; gsh_cascade_create).  k_lbp_tile's pair phase decides a stage with one lookup instead of count adds. */
struct LbpStage { unsigned first, count; float threshold; unsigned truth; };

struct LbpArgs {
  const unsigned *padded;       /* n frames of (iw+1)*(ih+1) u32 */
  size_t frame_stride;          /* (iw+1)*(ih+1) */
  unsigned S;                   /* iw + 1 */
  unsigned limit_bytes;         /* (frame_stride - 1) * 4 (GUARD clamp) */
  int step;
  unsigned nweaks, nstages, nsub;
  const LbpScale *scales;
  const LbpGeom *geom;          /* [nscales][nweaks], BYTE offsets into the padded table */
  const LbpWeak *weak;
  const LbpStage *stage;
  const int32_t *subsets;
  const uint32_t *truth;         /* stage truth tables (LbpStage::truth), ntruth words; may be null when ntruth = 0 */
  unsigned ntruth;
  unsigned long long *mask;     /* n frames x total_chunks*kChunkWords (pre-zeroed) */
  unsigned *chunk_count;        /* n frames x total_chunks (pre-zeroed) */
  unsigned total_chunks;
  /* Early exit at max_rects (ref :819-823).  Chunks are numbered in scan order (scale, then window
   * rows); 32 consecutive chunks form a group, 32 groups a super-group.  A finished chunk adds its
   * detections to its group's and its super-group's counter; a starting chunk sums the counters
   * of everything that lies wholly BEFORE its own group -- a lower bound of the detections that
   * precede it in the reference's scan order -- and skips when that reaches the cap. */
  unsigned *hits_group;         /* n frames x ngroups   (pre-zeroed) */
  unsigned *hits_super;         /* n frames x nsupers   (pre-zeroed) */
  unsigned *hits_total;         /* n frames (pre-zeroed): all detections published so far */
  unsigned scale0;              /* first scale of this launch (blockIdx.y counts from it) */
  unsigned ngroups, nsupers;
  unsigned nscales, cap;        /* cap = max_rects */
  unsigned nwindows_cap;        /* min(windows of the whole scan, 2^32 - 1): a cap >= this can never be reached */
  unsigned xcd_swizzle;         /* 1: chunk = (blockIdx.x % 8) * ceil(nchunks / 8) + blockIdx.x / 8 */
  unsigned long long *evaluated; /* optional (COUNT kernels): [0] += windows of every chunk that was not
                                    skipped, [1] += weak classifiers evaluated, summed over windows, [2] += dword
                                    corner loads issued, summed over lanes, [3] unused (round 3's stage prefilter
                                    counted here; the buffer keeps its four entries) */
};
constexpr unsigned kLbpGroupShift = 5, kLbpSuperShift = 10; /* chunks per group / super-group (log2) */
/* a detection counter as the L2 holds it (the L1 is per CU and not coherent): a stale value is harmless, it only skips less */
GS_DEV unsigned lbp_counter_load(const unsigned *p) {
#ifdef GS_EMU
  return *p;
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* global_load_dword sc1: device scope, no L1 hit */
#endif
}

/* cascade tables of one scale, staged in LDS by the block: every lane of every wave evaluates
 * the same weak classifier of the same scale, so these reads are same-address broadcasts
 * (low latency, no bank conflicts); only the subset word is a per-lane LDS read */
struct LbpLds {
  const LbpStage *stage;
  const LbpWeak *weak;
  const LbpGeom *geom;
  const int32_t *subsets;
};
GS_DEV size_t lbp_lds_bytes(unsigned nstages, unsigned nweaks, unsigned nsub) {
  return (size_t)nstages * sizeof(LbpStage) + (size_t)nweaks * (sizeof(LbpWeak) + sizeof(LbpGeom)) +
         (size_t)nsub * 4;
}
GS_DEV LbpLds lbp_stage_tables(char *smem, const LbpArgs &a, const LbpGeom *geom_scale,
                               unsigned tid, unsigned nthreads) {
  uint32_t *d = (uint32_t *)smem;
  const unsigned n0 = a.nstages * 4, n1 = a.nweaks * 4, n3 = a.nsub;
  for (unsigned i = tid; i < n0; i += nthreads) d[i] = ((const uint32_t *)a.stage)[i];
  for (unsigned i = tid; i < n1; i += nthreads) d[n0 + i] = ((const uint32_t *)a.weak)[i];
  for (unsigned i = tid; i < n1; i += nthreads) d[n0 + n1 + i] = ((const uint32_t *)geom_scale)[i];
  for (unsigned i = tid; i < n3; i += nthreads) d[n0 + 2 * n1 + i] = ((const uint32_t *)a.subsets)[i];
  LbpLds t;
  t.stage = (const LbpStage *)d;
  t.weak = (const LbpWeak *)(d + n0);
  t.geom = (const LbpGeom *)(d + n0 + n1);
  t.subsets = (const int32_t *)(d + n0 + 2 * n1);
  return t;
}

/* stages [s0, s1) of the cascade for the window whose top-left padded-table BYTE offset is
 * `origin`; returns false as soon as a stage sum falls below its threshold (ref :794-811).
 * The 16 corner loads are global loads at origin + (wave-uniform feature offset): the per-lane
 * part of the address is the window origin (constant for the whole cascade), the per-feature part
 * comes from the LDS-staged geometry table via v_readfirstlane.  (Buffer gathers with the uniform
 * part in an SGPR soffset measured slower.)  Cells are formed from column differences (21 instead
 * of 27 add/sub).  GUARD (feature rectangles that can stick out of the window, only with
 * scale < 1 after the clamp of ref :803-804): explicit clamped address instead; the reference
 * reads out of bounds there, so no particular value is "right". */
/* The 16 corner gathers of the NEXT weak classifier are issued before the current one's
 * arithmetic (they do not depend on it; only a stage end does), so L2 latency overlaps the ~70
 * lane-ops of cell/code/lookup work.  Tables are in evaluation order (stage by stage), so the
 * classifiers of stages [s0, s1) are the contiguous range [stage[s0].first, end of stage s1-1). */
struct LbpCorners { unsigned G[4][4]; };
template <bool GUARD>
GS_DEV LbpCorners lbp_gather(const LbpLds &t, const unsigned *Pg, unsigned origin, unsigned limit,
                             unsigned wi) {
  const LbpGeom g = t.geom[wi];
  const unsigned off0 = uniform((unsigned)g.off0), fw = uniform((unsigned)g.fw),
                 fhs = uniform((unsigned)g.fh_stride);
  LbpCorners c;
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      unsigned idx = origin + off0 + (unsigned)j * fhs + (unsigned)i * fw;
      if (GUARD) idx = idx > limit ? limit : idx;
      c.G[j][i] = *(const unsigned *)((const char *)Pg + idx);
    }
  return c;
}

#ifndef GS_LBP_PREFETCH
#define GS_LBP_PREFETCH 1 /* classifiers whose corners are in flight ahead of the arithmetic (MI355X: 1 -> 30.6, 2 -> 28.1, 3 -> 24.9 Gwin/s) */
#endif
template <bool GUARD, bool COUNT = false>
GS_DEV bool lbp_window_stages(const LbpLds &t, const unsigned *Pg, unsigned origin,
                              unsigned limit, unsigned s0, unsigned s1, unsigned *evals = nullptr) {
  constexpr int PD = GS_LBP_PREFETCH;
  const unsigned wend = uniform(t.stage[s1 - 1].first) + uniform(t.stage[s1 - 1].count);
  unsigned wi = uniform(t.stage[s0].first);
  LbpCorners q[PD];
#pragma unroll
  for (int d = 0; d < PD; d++) /* beyond wend: harmless re-read of the last classifier's corners */
    q[d] = lbp_gather<GUARD>(t, Pg, origin, limit, wi + d < wend ? wi + d : wend - 1);
  for (unsigned s = s0; s < s1; s++) {
    const LbpStage st = t.stage[s];
    const unsigned count = uniform(st.count);
    float sum = 0.0f;
    for (unsigned k = 0; k < count; k++, wi++) {
      const LbpCorners cur = q[0];
      if constexpr (COUNT) ++*evals;
#pragma unroll
      for (int d = 0; d + 1 < PD; d++) q[d] = q[d + 1];
      const LbpWeak wk = t.weak[wi];
      if (wi + PD < wend) q[PD - 1] = lbp_gather<GUARD>(t, Pg, origin, limit, wi + PD); /* wave-uniform */
      unsigned D[3][4], c[3][3];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) D[j][i] = cur.G[j + 1][i] - cur.G[j][i];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) c[j][i] = D[j][i + 1] - D[j][i];
      const unsigned ctr = c[1][1];
      const unsigned code = ((c[0][0] >= ctr) << 7) | ((c[0][1] >= ctr) << 6) |
                            ((c[0][2] >= ctr) << 5) | ((c[1][2] >= ctr) << 4) |
                            ((c[2][2] >= ctr) << 3) | ((c[2][1] >= ctr) << 2) |
                            ((c[2][0] >= ctr) << 1) | ((c[1][0] >= ctr) << 0);
      const unsigned word = code >> 5, bit = code & 31u;
      bool hit = false;
      if (word < wk.nsub) hit = ((uint32_t)t.subsets[wk.sub_off + word] >> bit) & 1u;
      sum += hit ? wk.left : wk.right;
    }
    if (sum < st.threshold) return false;
  }
  return true;
}

/* The same stages with FOUR lanes per window (a DPP quad): lane ci of the quad loads column ci of the 4 x 4 corner
 * grid, one gather instruction per grid ROW.  Why: the cascade is bound by the texture path (TA busy 95 %,
 * profiles/r03d_pmc_lbp.txt), which pays per 64-byte line a gather touches once the lanes scatter -- and re-packed
 * survivors do: one window per lane makes every gather visit 64 unrelated places (~27 TA cycles per gather in the
 * survivor phases against ~4 in the dense phase), although the four corners of one window's grid row lie only
 * 3 fw dwords apart, mostly inside one or two lines.  With a quad per window a gather covers 16 windows x one grid
 * row: the same 16 gather instructions per 64 windows and classifier, but 16 x (1..4) lines each instead of up
 * to 64.  The cells come from the neighbour lane through quad_perm (c = D[lane + 1] - D[lane]), the centre is
 * broadcast from lane 1, every lane contributes the code bits of its column and two quad ORs assemble the code;
 * the lookup and the float sum run redundantly in all four lanes, so the quad decides as one.
 * Control flow stays wave-uniform (a dead quad's loads are masked, its arithmetic runs on garbage): `live` says
 * whether this lane's window exists; returns the window's verdict in all four lanes. */
template <bool GUARD, bool COUNT = false>
GS_DEV bool lbp_quad_stages(const LbpLds &t, const unsigned *Pg, unsigned origin, unsigned ci, bool live,
                            unsigned limit, unsigned s0, unsigned s1, unsigned *evals = nullptr) {
  /* code bit of this lane's cell in the top / middle / bottom cell row (tl tc tr r br bc bl l = bits 7..0,
   * ref :780-782); lane 1's middle cell is the centre, lane 3 holds no cell */
  const unsigned sh0 = ci == 0u ? 7u : ci == 1u ? 6u : 5u, sh1 = ci == 0u ? 0u : 4u, sh2 = ci == 0u ? 1u : ci == 1u ? 2u : 3u;
  const unsigned use02 = ci < 3u ? 1u : 0u, use1 = (ci == 0u || ci == 2u) ? 1u : 0u;
  const unsigned wend = uniform(t.stage[s1 - 1].first) + uniform(t.stage[s1 - 1].count);
  unsigned wi = uniform(t.stage[s0].first);
  bool alive = live;
  auto gather = [&](unsigned w, unsigned (&G)[4]) {
    const LbpGeom g = t.geom[w];
    /* ci * fw without a 32-bit multiply (quarter rate): ci is 0..3 */
    const unsigned fw = uniform((unsigned)g.fw), cfw = ((ci & 1u) ? fw : 0u) + ((ci & 2u) ? 2u * fw : 0u);
    const unsigned col = origin + uniform((unsigned)g.off0) + cfw, fhs = uniform((unsigned)g.fh_stride);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      unsigned idx = col + (unsigned)j * fhs;
      if (GUARD) idx = idx > limit ? limit : idx;
      G[j] = *(const unsigned *)((const char *)Pg + idx);
    }
  };
  unsigned q[4] = {0, 0, 0, 0};
  if (alive) gather(wi, q);
  for (unsigned s = s0; s < s1; s++) {
    const LbpStage st = t.stage[s];
    const unsigned count = uniform(st.count);
    float sum = 0.0f;
    for (unsigned k = 0; k < count; k++, wi++) {
      const unsigned G0 = q[0], G1 = q[1], G2 = q[2], G3 = q[3];
      if constexpr (COUNT) {
        if (alive && ci == 0u) ++*evals;
      }
      const LbpWeak wk = t.weak[wi];
      if (wi + 1u < wend && alive) gather(wi + 1u, q); /* the next classifier's corners fly during this one's arithmetic */
      const unsigned D0 = G1 - G0, D1 = G2 - G1, D2 = G3 - G2;
      const unsigned c0 = quad_perm<1, 2, 3, 3>(D0) - D0, c1 = quad_perm<1, 2, 3, 3>(D1) - D1, c2 = quad_perm<1, 2, 3, 3>(D2) - D2;
      const unsigned ctr = quad_perm<1, 1, 1, 1>(c1);
      unsigned code = ((c0 >= ctr ? use02 : 0u) << sh0) | ((c1 >= ctr ? use1 : 0u) << sh1) | ((c2 >= ctr ? use02 : 0u) << sh2);
      code |= quad_perm<1, 0, 3, 2>(code);
      code |= quad_perm<2, 3, 0, 1>(code);
      /* subset bit, branch-free: words past the classifier's subset count read word 0 and are masked */
      const unsigned word = code >> 5, nsub = uniform(wk.nsub);
      const unsigned v = (uint32_t)t.subsets[uniform(wk.sub_off) + (word < nsub ? word : 0u)];
      const bool hit = word < nsub && ((v >> (code & 31u)) & 1u);
      sum += hit ? wk.left : wk.right;
    }
    if (sum < st.threshold) alive = false;
    if (!ballot(alive)) break; /* wave-uniform */
  }
  return alive;
}

GS_DEV unsigned lbp_origin(const LbpArgs &a, const LbpScale &sc, unsigned idx) {
  const unsigned yi = idx / sc.nx, xi = idx - yi * sc.nx;
  return ((yi * (unsigned)a.step) * a.S + xi * (unsigned)a.step) * 4u;
}

/* 63 % of windows die in stage 0 and ~99 % by stage 4, but a wave keeps executing while ANY of
 * its 64 lanes is alive: run densely, a wave executes ~35 weak classifiers for an average of ~6
 * per window.  So a block works through its chunk (2048 consecutive windows of one scale) in
 * PHASES of a few stages each: phase 0 evaluates every window, survivors are re-packed into an
 * LDS queue (one ds_add per wave), the next phase runs only over the queue, 64 survivors to a
 * wave, and so on until the last phase sets the window's bit in a 2048-bit LDS mask.  Survivors
 * of one chunk sit within a row or two of the integral image, so the re-packed gathers still hit
 * lines the dense phase just pulled into L1/L2 (a global survivor list loses exactly that and
 * measured 3-4x SLOWER).  The finished mask words and their popcount go to the ordered
 * compaction (k_compact.h) -- same bits as publishing ballots, no atomics on global memory. */
constexpr unsigned kLbpMaxPhases = 8;
struct LbpPhases { /* phase p = stages [end[p-1], end[p]) */
  unsigned n;
  unsigned end[kLbpMaxPhases];
  /* adaptive_max > 0: the FIRST re-packing point is chosen per block.  The dense phase (one window per lane,
   * consecutive windows: the cheapest gathers) runs stages [0, end[0]) and then goes on stage by stage, dead
   * windows masked, while more than adaptive_tenths/10 of the chunk's windows are alive and fewer than
   * adaptive_max stages are done: re-packing a set that has barely shrunk trades cheap consecutive gathers for
   * scattered ones and saves few lanes.  Block-noise frames (63 % die in stage 0) re-pack after stage 2 as the
   * fixed split (2,4,7) does (3.92 vs 3.90 ms per 4K frame); on edge maps, where far more windows of a chunk
   * survive the first stages, blocks stay dense for up to 6 stages: 6.33 -> 5.76 ms per 4K frame
   * (profiles/r02h_lbp_adaptive.log; fixed splits tuned for one input lose 3-10 % on the other,
   * profiles/r02h_lbp_splits.log).  The later re-packing points follow at +2 and +5 stages. */
  unsigned adaptive_max;
  unsigned adaptive_tenths; /* re-pack once alive <= tenths/10 of the chunk */
  unsigned adaptive_next[3]; /* later re-packing points, in stages after the first one (0 = none) */
  unsigned quad;             /* 1: re-packed survivors are evaluated four lanes per window (lbp_quad_stages) */
  /* k_lbp_tile (k_lbp_tile.h): a WAVE runs its windows densely through stages [0, tile_first) and then stage by stage while more
   * than tile_tenths/10 of them are alive (and fewer than adaptive_max stages are done); after that one lane per (window,
   * classifier) pair.  A pair costs about twice a dense lane-evaluation, so the switch pays from half the windows dead:
   * 1 and 7 (profiles/r05c_lbp_tile_v3_own_tables_addc.log: 4/10 3.67, 5/10 3.55, 6/10 3.53, 7/10 3.53 ms per 4K edge map; two dense
   * stages first: 3.70). */
  unsigned tile_first, tile_tenths;
};

/* grid (max chunks per scale, nscales, n frames), block 256;
 * dynamic LDS = lbp_lds_bytes(...) + 2 queues x 2048 u16 + 64 mask words + 2 counters */
GS_DEV size_t lbp_block_lds_bytes(unsigned nstages, unsigned nweaks, unsigned nsub) {
  return lbp_lds_bytes(nstages, nweaks, nsub) + 2 * kChunkItems * 2 + 64 * 4 + 16;
}

template <bool GUARD, bool COUNT = false>
__global__ __launch_bounds__(256) void k_lbp_cascade(LbpArgs a, LbpPhases ph) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  GS_DYN_LDS(smem);
  const unsigned si = blockIdx.y + a.scale0;
  const LbpScale sc = a.scales[si];
  /* XCD-aware chunk mapping (a.xcd_swizzle; gridDim.x is then a multiple of 8): the dispatcher places block b on
   * XCD b % 8, each with its own 4 MB L2.  Handing consecutive chunks to consecutive XCDs makes every XCD sweep
   * the whole integral image of every scale (one L2 request in three missed: 134 M misses per 4 x 1080p,
   * profiles/r02l_pmc_tcc.txt); instead XCD k takes the k-th eighth of the scale's chunks, i.e. one band of
   * window rows whose table rows stay in its L2.  Results do not depend on it (mask words are indexed by chunk). */
  unsigned cx = blockIdx.x;
  if (a.xcd_swizzle) {
    const unsigned per = (sc.nchunks + 7u) >> 3, j = blockIdx.x >> 3;
    if (j >= per) return; /* whole block */
    cx = (blockIdx.x & 7u) * per + j;
  }
  if (cx >= sc.nchunks) return; /* whole block */
  const unsigned tid = threadIdx.x;
  /* The reference stops scanning once max_rects detections exist (ref :819-823), and the output is
   * the FIRST max_rects hits in (scale, y, x) order.  Detections published by chunks that precede
   * this one in that order can only grow, so once they reach the cap no window of this chunk can
   * be among the first max_rects: skip the block (its mask words and counter stay zero).  The
   * counters are read at device scope (past the CU's L1, where the adds are performed); a stale or
   * partial sum only skips less, so the result is exact for any dispatch order -- and blocks are
   * dispatched scale by scale (inside a scale: in chunk order, or as eight bands side by side with the
   * XCD-aware mapping), so on frames that reach the cap every later scale is skipped, and most of the
   * scale in which it happens when the chunks run in order. */
  const unsigned lin = sc.chunk_base + cx;
  __shared__ unsigned before_s;
  if (tid < 64u) { /* one wave sums; the decision must be the same for the whole block */
    const unsigned g1 = lin >> kLbpGroupShift, g2 = lin >> kLbpSuperShift;
    unsigned *hs = a.hits_super + (size_t)blockIdx.z * a.nsupers;
    unsigned *hg = a.hits_group + (size_t)blockIdx.z * a.ngroups;
    unsigned before = 0;
    /* round 5: while the frame's total is below the cap nothing can be skipped -- one load instead of ~60-90 counters -- and
     * the counters are read with plain loads that bypass the L1 (the returning atomics of rounds 2-4 were serialised per
     * address by the L2; a stale value only skips less) */
    if (a.cap < a.nwindows_cap && lbp_counter_load(&a.hits_total[blockIdx.z]) >= a.cap) {
      for (unsigned q = tid; q < g2; q += 64u) before += lbp_counter_load(&hs[q]);
      const unsigned gq = (g2 << (kLbpSuperShift - kLbpGroupShift)) + tid; /* the <= 31 earlier groups of the own super-group */
      if (gq < g1) before += lbp_counter_load(&hg[gq]);
    }
    before = wave_sum(before);
    if (tid == 0) before_s = before;
  }
  __syncthreads();
  if (before_s >= a.cap) return;
  const LbpLds t = lbp_stage_tables(smem, a, a.geom + (size_t)si * a.nweaks, tid, 256u);
  char *extra = smem + ((lbp_lds_bytes(a.nstages, a.nweaks, a.nsub) + 15) & ~(size_t)15);
  uint16_t *queue = (uint16_t *)extra;                       /* [2][kChunkItems] */
  uint32_t *bits = (uint32_t *)(extra + 2 * kChunkItems * 2); /* [64] */
  unsigned *qn = (unsigned *)(bits + 64);                     /* [2] */
  if (tid < 64) bits[tid] = 0;
  if (tid < 2) qn[tid] = 0;
  __syncthreads();
  const unsigned nwin = sc.nx * sc.ny, first = cx * kChunkItems;
  const unsigned *Pg = a.padded + (size_t)blockIdx.z * a.frame_stride;
  unsigned n_in = nwin - first < kChunkItems ? nwin - first : kChunkItems;
  unsigned cur = 0, evals = 0;
  unsigned ends[kLbpMaxPhases], np = ph.n, pstart = 0;
#pragma unroll
  for (unsigned i = 0; i < kLbpMaxPhases; i++) ends[i] = ph.end[i];
  if (ph.adaptive_max) { /* dense pre-phase with a block-local choice of the first re-packing point */
    /* three rotating counters, one barrier per iteration: iteration i adds into slot i % 3 and reads it behind
     * the barrier; slot (i + 1) % 3 is cleared BEFORE that barrier -- its last readers (iteration i - 2) are all
     * past barrier i - 1, its next writers (iteration i + 1) all behind barrier i, so no add can be wiped */
    __shared__ unsigned alive_count[3];
    if (tid < 3) alive_count[tid] = 0;
    __syncthreads();
    unsigned slot = 0;
    unsigned alive = 0; /* bit k: window k * 256 + tid of the chunk is still alive */
#pragma unroll
    for (unsigned k = 0; k < kChunkItems / 256u; k++) alive |= (k * 256u + tid < n_in ? 1u : 0u) << k;
    /* stages [s_prev, e) are evaluated per iteration */
    unsigned s_prev = 0, e = ph.end[0] < a.nstages ? ph.end[0] : a.nstages;
    for (;;) { /* block-uniform */
      if (e > s_prev)
        for (unsigned k = 0; k < kChunkItems / 256u; k++) { /* uniform trip count; dead lanes idle */
          if (k * 256u >= n_in) break;
          if ((alive >> k) & 1u) {
            if (!lbp_window_stages<GUARD, COUNT>(t, Pg, lbp_origin(a, sc, first + k * 256u + tid), a.limit_bytes, s_prev, e, &evals))
              alive &= ~(1u << k);
          }
        }
      const unsigned mine = wave_sum((unsigned)__popc(alive));
      const unsigned nslot = slot == 2u ? 0u : slot + 1u;
      if ((tid & 63u) == 0 && mine) atomicAdd(&alive_count[slot], mine);
      if (tid == 0) alive_count[nslot] = 0;
      __syncthreads();
      const unsigned c = alive_count[slot];
      slot = nslot;
      if (e >= a.nstages || c == 0 || c * 10u <= ph.adaptive_tenths * n_in || e >= ph.adaptive_max) {
        if (e >= a.nstages) { /* small cascade: the dense phase was the whole cascade */
          for (unsigned k = 0; k < kChunkItems / 256u; k++)
            if ((alive >> k) & 1u) atomicOr(&bits[(k * 256u + tid) >> 5], 1u << ((k * 256u + tid) & 31u));
          n_in = 0;
        } else { /* re-pack the survivors: same queue format as the fixed phases */
          uint16_t *qout = queue + (cur ^ 1u) * kChunkItems;
          for (unsigned k = 0; k < kChunkItems / 256u; k++) {
            if (k * 256u >= n_in) break;
            const bool pass = (alive >> k) & 1u;
            const uint64_t m = ballot(pass);
            if (m) {
              const unsigned lane = lane_id();
              unsigned base = 0;
              if (lane == 0) base = atomicAdd(&qn[cur ^ 1u], (unsigned)__popcll(m));
              base = readlane0(base);
              if (pass) qout[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(k * 256u + tid);
            }
          }
          __syncthreads();
          n_in = qn[cur ^ 1u];
          if (tid == 0) qn[cur] = 0; /* nobody reads it any more; cleared BEFORE the barrier that releases the next phase's adds */
          __syncthreads();
          cur ^= 1u;
        }
        break;
      }
      s_prev = e, e = e + 1u;
    }
    /* the remaining phases: +2, +5 stages, then the rest */
    np = 0, ends[np++] = e;
#pragma unroll
    for (unsigned i = 0; i < 3; i++)
      if (ph.adaptive_next[i] && e + ph.adaptive_next[i] < a.nstages) ends[np++] = e + ph.adaptive_next[i];
    if (e < a.nstages) ends[np++] = a.nstages;
    pstart = 1;
    if (n_in == 0) pstart = np; /* nothing left (or already final) */
  }
  for (unsigned p = pstart; p < np; p++) {
    const unsigned s0 = p ? ends[p - 1] : 0u, s1 = ends[p];
    const bool lastp = p + 1 == np;
    const uint16_t *qin = queue + cur * kChunkItems;
    uint16_t *qout = queue + (cur ^ 1u) * kChunkItems;
    if (ph.quad && p > 0u) { /* re-packed survivors: a quad of lanes per window, 64 windows per block iteration */
      for (unsigned i0 = 0; i0 < n_in; i0 += 64u) { /* block-uniform trip count */
        const unsigned i = i0 + (tid >> 2), ci = tid & 3u;
        const bool live = i < n_in;
        if (i0 + (tid & ~63u) / 4u >= n_in) continue; /* this wave's 16 windows all lie past the queue's end (wave-uniform): the
                                                         tail stages leave a chunk a handful of survivors, one wave's worth */
        const unsigned local = qin[live ? i : n_in - 1u];
        const bool pass = lbp_quad_stages<GUARD, COUNT>(t, Pg, lbp_origin(a, sc, first + local), ci, live, a.limit_bytes, s0, s1, &evals);
        const bool lead = pass && ci == 0u;
        if (lastp) {
          if (lead) atomicOr(&bits[local >> 5], 1u << (local & 31u));
        } else {
          const uint64_t m = ballot(lead);
          if (m) {
            const unsigned lane = lane_id();
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&qn[cur ^ 1u], (unsigned)__popcll(m));
            base = readlane0(base);
            if (lead) qout[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)local;
          }
        }
      }
    } else
    for (unsigned i0 = 0; i0 < n_in; i0 += 256u) { /* block-uniform trip count */
      const unsigned i = i0 + tid;
      bool pass = false;
      unsigned local = 0;
      if (i < n_in) {
        local = p ? qin[i] : i;
        pass = lbp_window_stages<GUARD, COUNT>(t, Pg, lbp_origin(a, sc, first + local), a.limit_bytes, s0, s1, &evals);
      }
      if (lastp) {
        if (pass) atomicOr(&bits[local >> 5], 1u << (local & 31u));
      } else { /* re-pack survivors: one LDS atomic per wave */
        const uint64_t m = ballot(pass);
        if (m) {
          const unsigned lane = lane_id();
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(&qn[cur ^ 1u], (unsigned)__popcll(m));
          base = readlane0(base);
          if (pass) qout[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)local;
        }
      }
    }
    __syncthreads();
    if (!lastp) {
      n_in = qn[cur ^ 1u];
      /* this phase's input counter becomes the output counter of the next phase: cleared before the barrier that
       * lets the next phase's atomics start (cleared behind it, a fast wave's first add could be wiped) */
      if (tid == 0) qn[cur] = 0;
      __syncthreads();
      cur ^= 1u;
      if (n_in == 0) break; /* block-uniform */
    }
  }
  __syncthreads();
  /* publish this chunk: 32 words of 64 bits + their total */
  const size_t chunk = (size_t)blockIdx.z * a.total_chunks + sc.chunk_base + cx;
  if (tid < 64) { /* one wave */
    unsigned c = 0;
    if (tid < kChunkWords) {
      const unsigned long long wv = (unsigned long long)bits[2 * tid] | ((unsigned long long)bits[2 * tid + 1] << 32);
      a.mask[chunk * kChunkWords + tid] = wv;
      c = (unsigned)__popcll(wv);
    }
    c = wave_sum(c);
    if (tid == 0 && c) { /* hits are rare: few atomics */
      a.chunk_count[chunk] = c;
      if (a.cap < a.nwindows_cap) { /* a cap no scan can reach needs no early-exit bookkeeping (thousands of chunks
                                       adding to ONE counter serialise: +1.4 ms per 4K frame when every chunk has hits) */
        atomicAdd(&a.hits_group[(size_t)blockIdx.z * a.ngroups + (lin >> kLbpGroupShift)], c);
        atomicAdd(&a.hits_super[(size_t)blockIdx.z * a.nsupers + (lin >> kLbpSuperShift)], c);
        atomicAdd(&a.hits_total[blockIdx.z], c);
      }
    }
  }
  if constexpr (COUNT) { /* measurement build of the kernel (gsh_lbp_count_evaluated) */
    const unsigned ev = wave_sum(evals);
    if ((tid & 63u) == 0) atomicAdd(a.evaluated + 1, (unsigned long long)ev), atomicAdd(a.evaluated + 2, 16ull * ev);
    if (tid == 0)
      atomicAdd(a.evaluated, (unsigned long long)(nwin - first < kChunkItems ? nwin - first : kChunkItems));
  }
}

/* compaction functor: item -> gs_rect {x, y, win_w, win_h} (ref :825-828) */
struct LbpEmit {
  const LbpScale *scales;
  unsigned nscales;
  int step;
  unsigned *rects; /* n frames x max_rects x 4 u32 */
  unsigned max_rects;
  GS_DEV void operator()(unsigned frame, size_t item, unsigned r) const {
    const unsigned chunk = (unsigned)(item / kChunkItems);
    unsigned s = 0;
    while (s + 1 < nscales && scales[s + 1].chunk_base <= chunk) s++;
    const LbpScale sc = scales[s];
    const unsigned idx = (unsigned)(item - (size_t)sc.chunk_base * kChunkItems);
    const unsigned yi = idx / sc.nx, xi = idx - yi * sc.nx;
    unsigned *o = rects + ((size_t)frame * max_rects + r) * 4u;
    o[0] = xi * (unsigned)step, o[1] = yi * (unsigned)step;
    o[2] = (unsigned)sc.win_w, o[3] = (unsigned)sc.win_h;
  }
};

}  // namespace gs
#endif
