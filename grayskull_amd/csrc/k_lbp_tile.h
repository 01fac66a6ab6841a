/*
 * k_lbp_tile.h -- gs_lbp_detect (grayskull.h:790-835) with the integral-image corners served from an LDS tile.
 *
 * Why (round 5): k_lbp_cascade's survivor phases (stages >= 2, 3.0 of 4.4 ms per 4K edge map) sit on the texture
 * path, which pays per 64-byte LINE a gather touches -- ~6 lines per window and classifier once the windows are
 * re-packed, against 1 in the dense phase.  The LDS has no line granularity: a scattered ds_read_b32 costs its bank
 * conflicts only (measured: 0.7 extra cycles on top of 2 per wave-instruction, profiles/r05b_lbp_counters_tile_rule.txt),
 * and a dense one (64 consecutive dwords) 2 cycles against ~4 on the vector L1.
 *
 * A block owns a TILE of TW x TH window positions of one scale and copies the table region those windows can touch
 * -- ((TW-1) step + win_w + 1) x ((TH-1) step + win_h + 1) dwords of the zero-bordered table -- into LDS once.
 * After that one barrier the block's waves never meet again:
 *
 *   dense phase   a wave owns R = TW*TH/NT wave-rows (64 horizontally consecutive windows each, one per lane) and
 *                 runs them stage by stage (dead windows masked) while more than `tile_tenths`/10 of ITS windows
 *                 are alive and fewer than `adaptive_max` stages are done -- the per-block rule of k_lbp_cascade,
 *                 decided per wave;
 *   pair phase    the wave's survivors go to the wave's own LDS queue and every later stage is evaluated with one
 *                 LANE PER (window, weak classifier) pair: a stage of n classifiers takes floor(64 / n) windows per
 *                 wave iteration, lane l evaluating classifier l % n of window l / n.  One ballot collects the
 *                 64 subset-lookup bits and every lane of a group looks its window's n bits up in the stage's TRUTH
 *                 TABLE (LbpStage::truth: the verdict of the reference's sequential float32 sum, ref :796-810, for every
 *                 pattern of lookup results, built by the host with the same float adds), so the group decides as one,
 *                 survivors are compacted in place (the write index never passes the read index) and a stage costs ONE
 *                 pass over the wave's survivors instead of n.  That also removes the tail: a lone window that reaches
 *                 stage 19 costs 18 wave iterations, not 136.  Stages without a table (more than 12 classifiers) add
 *                 left / right in weak order like the reference; stages of more than 32 run a window per lane.
 *
 * Detections are rare (<= max_rects per frame matter): a window that passes the last stage sets its bit in the
 * frame's raster-order mask with one global atomicOr and bumps its chunk's counter, so k_compact.h sees exactly
 * what k_lbp_cascade would have published.  The max_rects early exit keeps its exact form: a tile is skipped when
 * the detections published by chunks that lie wholly before the tile's FIRST window reach the cap.
 *
 * Counters (8 x 4K edge maps, profiles/r05b_*): the LDS pipe is 72 % busy and the VALU 60-90 % (by the issue-rate table
 * of scripts/ubench_valu.cpp), the texture path 3 % -- both on-chip pipes near their limit, which is why the block's own
 * LDS tables are kept small (no leaf values: the truth tables replace them) and the code bits are assembled with
 * full-rate compare + add-with-carry pairs (push_ge_u32).
 *
 * Measured and not kept (profiles/r05g_lbp_tile_runs_of_stages_not_kept.log): passes that evaluate a RUN of stages at once when
 * a wave has few survivors left (the idle lanes take the following stages' classifiers speculatively, so a lone window
 * reaches the last stage in 3-4 passes instead of 18) -- 3.570 vs 3.574 ms per 4K edge map on the same box: the tail of
 * dependent passes is not what a block waits for.  What does matter is blocks per CU: the same shape drops from 0.23 to 0.34
 * ms per scale where its tile stops fitting twice (scale 2.36 -> 2.59 for 128 x 32 windows).
 *
 * Stage 0 by chains (round 5, measured and not kept; scripts/experiments/not_kept/k_lbp_tile_stage0_chains.h): windows one cell
 * height apart share three of the four rows of a classifier's corner grid, so a wave that walks down such a chain reads 4
 * corners per window instead of 16 -- and 12 corner reads MORE per dense evaluation cost 7-9 % of a scan (the GS_LBP_SENS hook
 * below, profiles/r05p_lbp_dense_sensitivity.log).  The cell height differs per classifier, so stage 0 had to leave the waves'
 * own phases: (classifier, column segment, window row) units listed in chain order, an equal run per wave, the 64 lookup bits
 * of a unit left as a word in the LDS, two more barriers, truth table per window afterwards.  Same rectangles, 7-10 % SLOWER
 * at every size (8 x 4K edge maps 3.66 vs 3.42 ms, 1080p block noise 0.69 vs 0.62; profiles/r05q_lbp_stage0_chains_not_kept.log):
 * a chain is a dependent sequence (shift the grid, read a row, evaluate) where the window-per-lane form requests the next
 * classifier's sixteen corners before it evaluates the current one, and the waves wait for each other twice.
 *
 * Not for GUARD geometries (feature rectangles that leave the window: scale < 1) -- those stay with k_lbp_cascade.
 */
#ifndef GS_K_LBP_TILE_H
#define GS_K_LBP_TILE_H
#include "k_lbp.h"

#ifndef GS_LBP_TILE_ODD_STRIDE
#define GS_LBP_TILE_ODD_STRIDE 1
#endif

namespace gs {

/* per (scale, classifier), re-based to the tile's row stride: BYTE offsets inside the tile; sub = sub_off | min(nsub, 8) << 16 */
struct LbpTileGeom { unsigned off0, fw, fh_stride, sub; };
/* per-stage lane layout of the pair phase */
struct LbpPairStage { unsigned wn, magic; }; /* windows per wave iteration (0: stage too long, window-parallel); lane / n = lane * magic >> 16 */

struct LbpTileTables {
  const LbpStage *stage;      /* LDS */
  const LbpTileGeom *geom;    /* LDS */
  const int32_t *subsets;     /* LDS */
  const uint32_t *truth;      /* LDS */
  const LbpPairStage *pst;    /* LDS */
  const LbpWeak *weak_global; /* leaf values of stages WITHOUT a truth table: wave-uniform loads from global memory */
};

GS_HD size_t lbp_tile_align16(size_t b) { return (b + 15) & ~(size_t)15; }
GS_HD unsigned lbp_tile_stride(unsigned tw, unsigned step, unsigned win_w) { return ((tw - 1u) * step + win_w + 1u) | (GS_LBP_TILE_ODD_STRIDE ? 1u : 0u); }
/* dynamic LDS of a block: stages | geometry | subsets | truth tables | pair-phase lane layouts | the waves' queues | the tile */
GS_HD size_t lbp_tile_lds_bytes(unsigned nstages, unsigned nweaks, unsigned nsub, unsigned ntruth, unsigned tile_windows, size_t tile_dwords) {
  return lbp_tile_align16((size_t)nstages * sizeof(LbpStage)) + lbp_tile_align16((size_t)nweaks * sizeof(LbpTileGeom)) +
         lbp_tile_align16((size_t)nsub * 4) + lbp_tile_align16((size_t)ntruth * 4) + lbp_tile_align16((size_t)nstages * sizeof(LbpPairStage)) +
         lbp_tile_align16((size_t)tile_windows * 2) + tile_dwords * 4 + 16;
}

/* the 8-bit LBP code of ref :769-783 from the 4 x 4 corner grid: tl tc tr r br bc bl l = bits 7..0, most significant first */
GS_DEV unsigned lbp_code_of(const unsigned (&G)[4][4]) {
  unsigned D[3][4], c[3][3];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) D[j][i] = G[j + 1][i] - G[j][i];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int i = 0; i < 3; i++) c[j][i] = D[j][i + 1] - D[j][i];
  const unsigned ctr = c[1][1];
  unsigned code = 0;
  code = push_ge_u32(code, c[0][0], ctr), code = push_ge_u32(code, c[0][1], ctr), code = push_ge_u32(code, c[0][2], ctr);
  code = push_ge_u32(code, c[1][2], ctr), code = push_ge_u32(code, c[2][2], ctr), code = push_ge_u32(code, c[2][1], ctr);
  code = push_ge_u32(code, c[2][0], ctr), code = push_ge_u32(code, c[1][0], ctr);
  return code;
}
/* the subset bit of a code (ref :785-788); sub = sub_off | nsub << 16 */
GS_DEV unsigned lbp_subset_bit(const int32_t *subsets, unsigned sub, unsigned code) {
  const unsigned word = code >> 5, nsub = sub >> 16;
  const unsigned v = (uint32_t)subsets[(sub & 0xffffu) + (word < nsub ? word : 0u)];
  return word < nsub ? (v >> (code & 31u)) & 1u : 0u;
}

/* one (window, classifier) pair, geometry per lane */
GS_DEV bool lbp_pair_hit(const LbpTileTables &t, const unsigned *tile, unsigned origin, unsigned wi) {
  const LbpTileGeom g = t.geom[wi];
  const unsigned base = origin + g.off0;
  unsigned G[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) G[j][i] = *(const unsigned *)((const char *)tile + (base + (unsigned)j * g.fh_stride + (unsigned)i * g.fw));
  return lbp_subset_bit(t.subsets, g.sub, lbp_code_of(G)) != 0u;
}

/* stages [s0, s1) for ONE window per lane, every lane on the same classifier (geometry wave-uniform, ds_read_b32 of 64
 * consecutive dwords: conflict-free); the next classifier's corners are requested before the current one's arithmetic, as
 * in lbp_window_stages.  A stage's verdict comes from its truth table (bit k of the pattern = classifier k's lookup). */
template <bool COUNT>
GS_DEV bool lbp_tile_window_stages(const LbpTileTables &t, const unsigned *tile, unsigned origin, unsigned s0, unsigned s1,
                                   unsigned *evals) {
  if (s1 <= s0) return true; /* wave-uniform; an empty run of stages rejects nothing (and stage[s1 - 1] would be stage[-1]) */
  const unsigned wend = uniform(t.stage[s1 - 1].first) + uniform(t.stage[s1 - 1].count);
  unsigned wi = uniform(t.stage[s0].first);
#ifdef GS_LBP_SENS
  unsigned sens = 0; /* experiment hook: what 12 more corner reads (1) / ~16 more VALU operations (2) per dense evaluation cost */
#endif
  auto gather = [&](unsigned w, unsigned (&G)[4][4]) {
    const LbpTileGeom g = t.geom[w];
    const unsigned base = origin + uniform(g.off0), fw = uniform(g.fw), fhs = uniform(g.fh_stride);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) G[j][i] = *(const unsigned *)((const char *)tile + (base + (unsigned)j * fhs + (unsigned)i * fw));
#ifdef GS_LBP_SENS
#if GS_LBP_SENS == 1
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) sens += *(const unsigned *)((const char *)tile + (base + 4u + (unsigned)j * fhs + (unsigned)i * fw));
#else
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) sens = (sens ^ G[j][i]) + (sens >> 3);
#endif
#endif
  };
  unsigned Q[4][4];
  gather(wi, Q);
  for (unsigned s = s0; s < s1; s++) {
    const LbpStage st = t.stage[s];
    const unsigned count = uniform(st.count), tt = uniform(st.truth);
    unsigned pat = 0;
    float sum = 0.0f;
    for (unsigned k = 0; k < count; k++, wi++) {
      unsigned G[4][4];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) G[j][i] = Q[j][i];
      if constexpr (COUNT) ++*evals;
      const unsigned sub = uniform(t.geom[wi].sub);
      if (wi + 1u < wend) gather(wi + 1u, Q); /* wave-uniform */
      const unsigned bit = lbp_subset_bit(t.subsets, sub, lbp_code_of(G));
      if (tt) {
        pat |= bit << k;
      } else { /* no table for this stage: the reference's adds, leaf values from global memory (wave-uniform) */
        const LbpWeak wk = t.weak_global[wi];
        sum += bit ? wk.left : wk.right;
      }
    }
    const bool pass = tt ? ((t.truth[tt - 1u + (pat >> 5)] >> (pat & 31u)) & 1u) != 0u : !(sum < st.threshold);
    if (!pass) return false;
  }
#ifdef GS_LBP_SENS
  if (sens == 0x9e3779b9u) return false; /* keeps the extra work alive; never true in practice */
#endif
  return true;
}

struct LbpTilePos { unsigned x0w, y0w, nwx, nwy; }; /* the tile's first window (in window indices) and its extent */

/* The windows of the lanes with `lead` (top-left corner at dword `odw` of the tile) passed the last stage.  Called by the
 * whole wave.  Detections are rare with a real cascade, but a permissive one can pass a tenth of all windows, and then one
 * atomic per detection on its chunk's counter serialises in the L2 (a truncated frontalface cascade: 8.4 instead of 1.9 ms
 * per 4K edge map, profiles/r05y_lbp_stage_costs.log).  So the wave aggregates: one atomicOr per distinct mask word (the
 * lanes' bits are distinct, so their OR is their sum) and one add per distinct chunk. */
GS_DEV void lbp_tile_publish_wave(const LbpArgs &a, const LbpScale &sc, const LbpTilePos &tp, bool lead, unsigned odw, unsigned TS) {
  uint64_t todo = ballot(lead);
  if (!todo) return; /* wave-uniform */
  const unsigned ry = odw / TS, rx = odw - ry * TS, ly = ry / (unsigned)a.step, lx = rx / (unsigned)a.step; /* rare: a division is fine */
  const unsigned idx = (tp.y0w + ly) * sc.nx + tp.x0w + lx; /* raster index inside the scale */
  const unsigned lin = sc.chunk_base + idx / kChunkItems, bit = idx & (kChunkItems - 1u);
  const unsigned wid = lin * kChunkWords + (bit >> 6); /* mask word inside the frame's array */
  const uint32_t blo = (bit & 32u) ? 0u : 1u << (bit & 31u), bhi = (bit & 32u) ? 1u << (bit & 31u) : 0u;
  const unsigned lane = lane_id();
  const size_t frame_chunk0 = (size_t)blockIdx.z * a.total_chunks;
  while (todo) { /* one distinct mask word per trip */
    const unsigned first = (unsigned)__builtin_ctzll(todo);
    const unsigned w0 = readlane_at(wid, first);
    const bool mine = lead && wid == w0;
    const uint64_t same = ballot(mine);
    const uint32_t lo = wave_sum(mine ? blo : 0u), hi = wave_sum(mine ? bhi : 0u);
    if (lane == first) atomicOr(&a.mask[frame_chunk0 * kChunkWords + w0], (unsigned long long)lo | ((unsigned long long)hi << 32));
    todo &= ~same;
  }
  todo = ballot(lead);
  const unsigned total = (unsigned)__popcll(todo);
  while (todo) { /* one distinct chunk per trip */
    const unsigned first = (unsigned)__builtin_ctzll(todo);
    const unsigned l0 = readlane_at(lin, first);
    const uint64_t same = ballot(lead && lin == l0);
    if (lane == first) {
      const unsigned c = (unsigned)__popcll(same);
      atomicAdd(&a.chunk_count[frame_chunk0 + l0], c);
      if (a.cap < a.nwindows_cap) {
        atomicAdd(&a.hits_group[(size_t)blockIdx.z * a.ngroups + (l0 >> kLbpGroupShift)], c);
        atomicAdd(&a.hits_super[(size_t)blockIdx.z * a.nsupers + (l0 >> kLbpSuperShift)], c);
      }
    }
    todo &= ~same;
  }
  if (a.cap < a.nwindows_cap && lane == 0) atomicAdd(&a.hits_total[blockIdx.z], total);
}

/* grid (max tiles per scale [rounded up to 8 with the XCD mapping], scales of this launch, n frames), block NT;
 * dynamic LDS = lbp_tile_lds_bytes(...) for the LARGEST scale of the launch.
 * 1024-thread blocks are held to 64 registers (two blocks = 32 waves per CU is what the rule picks them for = 8 waves per
 * SIMD, the bound's second argument; 69-72 without it). */
template <unsigned NT, unsigned TW, unsigned TH, bool COUNT = false>
__global__ __launch_bounds__(NT, NT == 1024 ? 8 : 1) void k_lbp_tile(LbpArgs a, LbpPhases ph) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  static_assert(TW % 64u == 0 && (TW * TH) % NT == 0 && TW * TH <= 65536u, "tile shape");
  constexpr unsigned NW = NT / 64u, R = TW * TH / NT, SEG = TW / 64u; /* waves, wave-rows per wave, wave-rows per tile row */
  GS_DYN_LDS(smem);
  const unsigned si = blockIdx.y + a.scale0;
  const LbpScale sc = a.scales[si];
  const unsigned tiles_x = (sc.nx + TW - 1u) / TW, tiles_y = (sc.ny + TH - 1u) / TH, ntiles = tiles_x * tiles_y;
  unsigned tix = blockIdx.x;
  if (a.xcd_swizzle == 1u) { /* XCD k takes the k-th eighth of the scale's tiles: a band of tile rows that stays in its L2 (k_lbp.h) */
    const unsigned per = (ntiles + 7u) >> 3, j = blockIdx.x >> 3;
    if (j >= per) return;
    tix = (blockIdx.x & 7u) * per + j;
  } else if (a.xcd_swizzle >= 16u) { /* experiments: runs of G = xcd_swizzle - 16 tiles dealt round the XCDs */
    const unsigned G = a.xcd_swizzle - 16u, j = blockIdx.x >> 3;
    tix = ((j / G) * 8u + (blockIdx.x & 7u)) * G + j % G;
  }
  if (tix >= ntiles) return; /* whole block */
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  LbpTilePos tp;
  {
    const unsigned ty = tix / tiles_x, tx = tix - ty * tiles_x;
    tp.x0w = tx * TW, tp.y0w = ty * TH;
    tp.nwx = sc.nx - tp.x0w < TW ? sc.nx - tp.x0w : TW;
    tp.nwy = sc.ny - tp.y0w < TH ? sc.ny - tp.y0w : TH;
  }
  /* max_rects early exit (k_lbp.h): everything that lies wholly before the group of the tile's FIRST window precedes
   * every window of the tile in the reference's scan order.  The counters are requested here and summed behind the
   * staging below, so their round trip to the L2 (returning atomics) hides under the table / tile copies; a skipped block
   * has staged for nothing, but blocks are only skipped once a frame has reached the cap. */
  __shared__ unsigned before_s;
  const bool capped = a.cap < a.nwindows_cap;
  unsigned before_part = 0;
  if (capped && tid < 64u) {
    const unsigned lin = sc.chunk_base + (tp.y0w * sc.nx + tp.x0w) / kChunkItems;
    const unsigned g1 = lin >> kLbpGroupShift, g2 = lin >> kLbpSuperShift;
    unsigned *hs = a.hits_super + (size_t)blockIdx.z * a.nsupers;
    unsigned *hg = a.hits_group + (size_t)blockIdx.z * a.ngroups;
    /* everything published so far bounds what lies before this tile: while the frame's total is below the cap nothing can
     * be skipped and ONE load settles it (round 5: every block used to read ~60-90 counters with returning atomics, which
     * the L2 serialises per address -- the late scales, whose chunk numbers are large, paid for it).  Plain loads that
     * bypass the L1 do: a stale value only skips less. */
    if (lbp_counter_load(&a.hits_total[blockIdx.z]) >= a.cap) {
      for (unsigned q = tid; q < g2; q += 64u) before_part += lbp_counter_load(&hs[q]);
      const unsigned gq = (g2 << (kLbpSuperShift - kLbpGroupShift)) + tid;
      if (gq < g1) before_part += lbp_counter_load(&hg[gq]);
    }
  }
  /* ---- LDS: stages | geometry re-based to the tile's row stride | subsets | truth tables | lane layouts | queues | tile */
  const unsigned step = (unsigned)a.step;
  /* row stride of the tile in dwords: the columns the windows can touch, made ODD -- survivors cluster spatially, and with a
   * stride that shares a factor with the 32 banks vertically neighbouring windows would meet in the same banks */
  const unsigned TS = lbp_tile_stride(TW, step, (unsigned)sc.win_w), TR = (TH - 1u) * step + (unsigned)sc.win_h + 1u;
  char *p = smem;
  LbpStage *l_stage = (LbpStage *)p;
  p += lbp_tile_align16((size_t)a.nstages * sizeof(LbpStage));
  LbpTileGeom *l_geom = (LbpTileGeom *)p;
  p += lbp_tile_align16((size_t)a.nweaks * sizeof(LbpTileGeom));
  int32_t *l_sub = (int32_t *)p;
  p += lbp_tile_align16((size_t)a.nsub * 4);
  uint32_t *l_truth = (uint32_t *)p;
  p += lbp_tile_align16((size_t)a.ntruth * 4);
  LbpPairStage *l_pst = (LbpPairStage *)p;
  p += lbp_tile_align16((size_t)a.nstages * sizeof(LbpPairStage));
  uint16_t *queue_all = (uint16_t *)p; /* [NW][R * 64]: dword offsets of the survivors' top-left corners inside the tile */
  p += lbp_tile_align16((size_t)TW * TH * 2);
  unsigned *tile = (unsigned *)p;
  for (unsigned s = tid; s < a.nstages; s += NT) {
    const LbpStage st = a.stage[s];
    l_stage[s] = st;
    LbpPairStage q;
    q.wn = (st.count >= 1u && st.count <= 32u) ? 64u / st.count : 0u;
    q.magic = q.wn ? 65536u / st.count + 1u : 0u;
    l_pst[s] = q;
  }
  for (unsigned i = tid; i < a.nsub; i += NT) l_sub[i] = a.subsets[i];
  for (unsigned i = tid; i < a.ntruth; i += NT) l_truth[i] = a.truth[i];
  { /* the host's geometry holds byte offsets for the table's row stride a.S: re-base to TS (fy = off0 / (4 S), fx the rest) */
    const LbpGeom *gs_ = a.geom + (size_t)si * a.nweaks;
    for (unsigned i = tid; i < a.nweaks; i += NT) {
      const LbpGeom g = gs_[i];
      const LbpWeak wk = a.weak[i];
      const unsigned o = (unsigned)g.off0 >> 2, fy = o / a.S, fx = o - fy * a.S;
      LbpTileGeom r;
      r.off0 = (fy * TS + fx) * 4u, r.fw = (unsigned)g.fw, r.fh_stride = (unsigned)g.pad * TS * 4u;
      r.sub = (wk.sub_off & 0xffffu) | ((wk.nsub < 8u ? wk.nsub : 8u) << 16); /* a code has 8 bits: words >= 8 never match */
      l_geom[i] = r;
    }
  }
  {
    const unsigned *Pg = a.padded + (size_t)blockIdx.z * a.frame_stride;
    const unsigned rows = (unsigned)(a.frame_stride / a.S);
    const unsigned gx0 = tp.x0w * step, gy0 = tp.y0w * step;
    /* (Round 5, measured and not kept: four trips at a time with sixteen unconditional buffer loads in flight -- hipcc puts each
     * trip's four bounds-tested loads behind branches and waits for them before the next trip's are issued, up to eight memory
     * latencies in a row per block.  1-3 % SLOWER at every size: 8 x 4K edge maps 3.23 vs 3.19 ms, 1080p block noise 0.64 vs
     * 0.62, per scale +0.5-1 %; profiles/r05m_lbp_staging_16_loads_not_kept.log.  The other blocks of the CU cover the
     * staging as it is.) */
    for (unsigned r0 = wave * 4u; r0 < TR; r0 += NW * 4u) { /* four table rows per wave and trip: loads first, stores after */
      for (unsigned c = lane; c < TS; c += 64u) {
        unsigned v[4];
#pragma unroll
        for (unsigned k = 0; k < 4u; k++) {
          const unsigned gy = gy0 + r0 + k, gx = gx0 + c;
          v[k] = (r0 + k < TR && gy < rows && gx < a.S) ? Pg[(size_t)gy * a.S + gx] : 0u;
        }
#pragma unroll
        for (unsigned k = 0; k < 4u; k++)
          if (r0 + k < TR) tile[(r0 + k) * TS + c] = v[k];
      }
    }
  }
  if (capped && tid < 64u) {
    const unsigned before = wave_sum(before_part);
    if (tid == 0) before_s = before;
  }
  __syncthreads();
  if (capped && before_s >= a.cap) return; /* whole block */
  /* ---- from here on the waves are on their own */
  const LbpTileTables t{l_stage, l_geom, l_sub, l_truth, l_pst, a.weak};
  uint16_t *queue = queue_all + wave * (R * 64u);
  unsigned evals = 0;
  /* wave-row k of this wave: q = wave + NW k (interleaved, so every wave gets rows from all over the tile); the dword offset
   * of its window's top-left corner inside the tile */
  unsigned alive = 0; /* bit k: window (wave-row k, lane) is alive */
  auto odw_of = [&](unsigned k) {
    const unsigned q = wave + NW * k, ly = q / SEG, lx = (q - ly * SEG) * 64u + lane;
    return ly * step * TS + lx * step;
  };
#pragma unroll
  for (unsigned k = 0; k < R; k++) {
    const unsigned q = wave + NW * k, ly = q / SEG, lx = (q - ly * SEG) * 64u + lane;
    alive |= ((lx < tp.nwx && ly < tp.nwy) ? 1u : 0u) << k;
  }
  const unsigned nvalid = wave_sum((unsigned)__popc(alive));
  unsigned m = 0; /* survivors in the wave's queue */
  unsigned e = 0;
  if (nvalid) {
    const unsigned emax = ph.adaptive_max ? ph.adaptive_max : ph.end[0];
    unsigned s_prev = 0;
    e = ph.tile_first < a.nstages ? ph.tile_first : a.nstages;
    for (;;) { /* wave-uniform */
#pragma clang loop unroll(disable) /* one copy of the stage loop, not R: registers (64 for the 1024-thread shapes) and code size */
      for (unsigned k = 0; k < R; k++) {
        if ((alive >> k) & 1u) {
          if (!lbp_tile_window_stages<COUNT>(t, tile, odw_of(k) * 4u, s_prev, e, &evals)) alive &= ~(1u << k);
        }
      }
      const unsigned c = wave_sum((unsigned)__popc(alive));
      if (e >= a.nstages || c == 0u || c * 10u <= ph.tile_tenths * nvalid || e >= emax) {
        m = c;
        break;
      }
      s_prev = e, e = e + 1u;
    }
    if (m) {
      if (e >= a.nstages) { /* short cascade: the dense phase was all of it */
#pragma unroll
        for (unsigned k = 0; k < R; k++) lbp_tile_publish_wave(a, sc, tp, (alive >> k) & 1u, odw_of(k), TS);
        m = 0;
      } else { /* re-pack the wave's survivors */
        unsigned out = 0;
#pragma unroll
        for (unsigned k = 0; k < R; k++) {
          const bool pass = (alive >> k) & 1u;
          const uint64_t mk = ballot(pass);
          if (mk) {
            if (pass) queue[out + mbcnt(mk)] = (uint16_t)odw_of(k);
            out += (unsigned)__popcll(mk);
          }
        }
      }
    }
  }
  /* ---- pair phase: one stage per pass over the queue */
  for (unsigned s = e; s < a.nstages && m; s++) { /* wave-uniform */
    const LbpStage st = t.stage[s];
    const unsigned n = uniform(st.count), first = uniform(st.first), tt = uniform(st.truth);
    const bool lastst = s + 1u == a.nstages;
    const LbpPairStage ps = t.pst[s];
    const unsigned wn = uniform(ps.wn);
    unsigned out = 0;
    wave_sync(); /* the queue as the previous stage (or the re-packing) left it */
    if (wn) {
      const unsigned g = (lane * uniform(ps.magic)) >> 16, k = lane - g * n;
      const bool act = g < wn;
      const unsigned sh = act ? g * n : 0u, pmask = n >= 32u ? 0xffffffffu : (1u << n) - 1u; /* n = 32: wn = 2, shifts 0 / 32 */
      for (unsigned b = 0; b < m; b += wn) { /* wave-uniform */
        const unsigned j = b + g;
        const bool live = act && j < m;
        const unsigned odw = queue[live ? j : 0u];
        bool hit = false;
        if (live) {
          hit = lbp_pair_hit(t, tile, odw * 4u, first + k);
          if constexpr (COUNT) ++evals;
        }
        const uint64_t hm = ballot(hit);
        const uint32_t mine = (uint32_t)(hm >> sh) & pmask;
        bool pass;
        if (tt) { /* the stage's verdict for this pattern of lookup results, precomputed with the reference's float adds */
          pass = (t.truth[tt - 1u + (mine >> 5)] >> (mine & 31u)) & 1u;
        } else {
          float sum = 0.0f;
          for (unsigned i = 0; i < n; i++) { /* the reference's order of adds; the leaf values are wave-uniform loads */
            const LbpWeak wk = t.weak_global[first + i];
            sum += ((mine >> i) & 1u) ? wk.left : wk.right;
          }
          pass = !(sum < st.threshold);
        }
        const bool lead = live && k == 0u && pass;
        const uint64_t pm = ballot(lead);
        if (pm) {
          if (lastst) {
            lbp_tile_publish_wave(a, sc, tp, lead, odw, TS);
          } else {
            if (lead) queue[out + mbcnt(pm)] = (uint16_t)odw; /* in place: out <= b, and this trip's reads are done (the ballot) */
            out += (unsigned)__popcll(pm);
          }
        }
      }
    } else { /* a stage of more than 32 classifiers: a window per lane */
      for (unsigned b = 0; b < m; b += 64u) {
        const unsigned j = b + lane;
        const bool live = j < m;
        const unsigned odw = queue[live ? j : 0u];
        bool pass = false;
        if (live) pass = lbp_tile_window_stages<COUNT>(t, tile, odw * 4u, s, s + 1u, &evals);
        const uint64_t pm = ballot(pass);
        if (pm) {
          if (lastst) {
            lbp_tile_publish_wave(a, sc, tp, pass, odw, TS);
          } else {
            if (pass) queue[out + mbcnt(pm)] = (uint16_t)odw;
            out += (unsigned)__popcll(pm);
          }
        }
      }
    }
    m = out;
  }
  if constexpr (COUNT) {
    const unsigned ev = wave_sum(evals);
    if (lane == 0) atomicAdd(a.evaluated + 1, (unsigned long long)ev), atomicAdd(a.evaluated + 2, 16ull * ev);
    if (tid == 0) atomicAdd(a.evaluated, (unsigned long long)(tp.nwx * tp.nwy));
  }
}

}  // namespace gs
#endif
