/*
 * k_lbp_tile.h -- gs_lbp_detect (grayskull.h:790-835) with the integral-image corners served from an LDS tile.
 *
 * Why (round 5): k_lbp_cascade's survivor phases (stages >= 2, 3.0 of 4.4 ms per 4K edge map) sit on the texture
 * path, which pays per 64-byte LINE a gather touches -- ~6 lines per window and classifier once the windows are
 * re-packed, against 1 in the dense phase.  The LDS has no line granularity: a scattered ds_read_b32 costs its bank
 * conflicts only (~3.5-way for 32 random dwords over 32 banks = 7 cycles per wave-instruction instead of 2), and a
 * dense one (64 consecutive dwords) 2 cycles against ~4 on the vector L1.
 *
 * A block owns a TILE of TW x TH window positions of one scale and copies the table region those windows can touch
 * -- ((TW-1) step + win_w + 1) x ((TH-1) step + win_h + 1) dwords of the zero-bordered table -- into LDS once.
 * After that one barrier the block's waves never meet again:
 *
 *   dense phase   a wave owns R = TW*TH/NT wave-rows (64 horizontally consecutive windows each, one per lane) and
 *                 runs them stage by stage (dead windows masked) while more than `adaptive_tenths`/10 of ITS windows
 *                 are alive and fewer than `adaptive_max` stages are done -- the per-block rule of k_lbp_cascade,
 *                 decided per wave;
 *   pair phase    the wave's survivors go to the wave's own LDS queue and every later stage is evaluated with one
 *                 LANE PER (window, weak classifier) pair: a stage of n classifiers takes floor(64 / n) windows per
 *                 wave iteration, lane l evaluating classifier l % n of window l / n.  One ballot collects the
 *                 64 subset-lookup bits; every lane then forms its window's stage sum as the reference does --
 *                 sequential float32 adds of left/right in weak order (ref :796-810), the leaf values read
 *                 wave-uniformly -- so the group decides as one, survivors are compacted in place (the write index
 *                 never passes the read index) and a stage costs ONE pass over the wave's survivors instead of n.
 *                 That also removes the tail: a lone window that reaches stage 19 costs 18 wave iterations, not 136.
 *
 * Detections are rare (<= max_rects per frame matter): a window that passes the last stage sets its bit in the
 * frame's raster-order mask with one global atomicOr and bumps its chunk's counter, so k_compact.h sees exactly
 * what k_lbp_cascade would have published.  The max_rects early exit keeps its exact form: a tile is skipped when
 * the detections published by chunks that lie wholly before the tile's FIRST window reach the cap.
 *
 * Not for GUARD geometries (feature rectangles that leave the window: scale < 1) -- those stay with k_lbp_cascade.
 */
#ifndef GS_K_LBP_TILE_H
#define GS_K_LBP_TILE_H
#include "k_lbp.h"

namespace gs {

/* per-stage lane layout of the pair phase, built once per block next to the cascade tables */
struct LbpPairStage { unsigned wn, magic; }; /* windows per wave iteration (0: stage too long, window-parallel); lane / n = lane * magic >> 16 */

/* dynamic LDS of a block: cascade tables | pair-phase lane layouts | stage truth tables | the waves' queues | the tile */
GS_HD size_t lbp_tile_lds_bytes(unsigned nstages, unsigned nweaks, unsigned nsub, unsigned ntruth, unsigned tile_windows, size_t tile_dwords) {
  const size_t tables = (size_t)nstages * sizeof(LbpStage) + (size_t)nweaks * (sizeof(LbpWeak) + sizeof(LbpGeom)) + (size_t)nsub * 4;
  return ((tables + 15) & ~(size_t)15) + (((size_t)nstages * sizeof(LbpPairStage) + 15) & ~(size_t)15) + (((size_t)ntruth * 4 + 15) & ~(size_t)15) +
         (((size_t)tile_windows * 2 + 15) & ~(size_t)15) + tile_dwords * 4 + 16;
}

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
  return ((c[0][0] >= ctr) << 7) | ((c[0][1] >= ctr) << 6) | ((c[0][2] >= ctr) << 5) | ((c[1][2] >= ctr) << 4) |
         ((c[2][2] >= ctr) << 3) | ((c[2][1] >= ctr) << 2) | ((c[2][0] >= ctr) << 1) | ((c[1][0] >= ctr) << 0);
}

/* one (window, classifier) pair, geometry per lane: the subset bit of the window's LBP code (ref :769-788).  The tile
 * kernel re-bases the staged geometry to the tile's row stride and packs the classifier's subset range into the spare
 * word (pad = sub_off | min(nsub, 8) << 16), so a pair costs one 16-byte table read. */
GS_DEV bool lbp_pair_hit(const LbpLds &t, const unsigned *tile, unsigned origin, unsigned wi) {
  const LbpGeom g = t.geom[wi];
  const unsigned base = origin + (unsigned)g.off0, sub_off = (unsigned)g.pad & 0xffffu, nsub = (unsigned)g.pad >> 16;
  unsigned G[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++)
      G[j][i] = *(const unsigned *)((const char *)tile + (base + (unsigned)j * (unsigned)g.fh_stride + (unsigned)i * (unsigned)g.fw));
  const unsigned code = opaque(lbp_code_of(G)); /* opaque: all 16 corners are fetched in ONE LDS round trip (the optimiser otherwise tests
                                                    word < nsub on the code's top bits first and sinks six loads behind that branch) */
  const unsigned word = code >> 5;
  const unsigned v = (uint32_t)t.subsets[sub_off + (word < nsub ? word : 0u)];
  return word < nsub && ((v >> (code & 31u)) & 1u);
}

struct LbpTilePos { unsigned x0w, y0w, nwx, nwy; }; /* the tile's first window (in window indices) and its extent */

/* the window whose top-left corner sits at dword `odw` of the tile passed the last stage (rare: a division is fine) */
GS_DEV void lbp_tile_publish(const LbpArgs &a, const LbpScale &sc, const LbpTilePos &tp, unsigned odw, unsigned TS) {
  const unsigned ry = odw / TS, rx = odw - ry * TS, ly = ry / (unsigned)a.step, lx = rx / (unsigned)a.step;
  const unsigned idx = (tp.y0w + ly) * sc.nx + tp.x0w + lx; /* raster index inside the scale */
  const unsigned lin = sc.chunk_base + idx / kChunkItems, bit = idx & (kChunkItems - 1u);
  const size_t chunk = (size_t)blockIdx.z * a.total_chunks + lin;
  atomicOr(&a.mask[chunk * kChunkWords + (bit >> 6)], 1ull << (bit & 63u));
  atomicAdd(&a.chunk_count[chunk], 1u);
  if (a.cap < a.nwindows_cap) {
    atomicAdd(&a.hits_group[(size_t)blockIdx.z * a.ngroups + (lin >> kLbpGroupShift)], 1u);
    atomicAdd(&a.hits_super[(size_t)blockIdx.z * a.nsupers + (lin >> kLbpSuperShift)], 1u);
    atomicAdd(&a.hits_total[blockIdx.z], 1u);
  }
}

/* grid (max tiles per scale [rounded up to 8 with the XCD mapping], scales of this launch, n frames), block NT;
 * dynamic LDS = lbp_tile_lds_bytes(...) for the LARGEST scale of the launch */
/* 1024-thread blocks are held to 64 registers (two blocks = 32 waves per CU is what the rule picks them for = 8 waves per SIMD, the bound's second argument; 69-72 without
 * the bound, no scratch with it) */
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
  if (a.xcd_swizzle) { /* XCD k takes the k-th eighth of the scale's tiles: a band of tile rows that stays in its L2 (k_lbp.h) */
    const unsigned per = (ntiles + 7u) >> 3, j = blockIdx.x >> 3;
    if (j >= per) return;
    tix = (blockIdx.x & 7u) * per + j;
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
   * every window of the tile in the reference's scan order */
  __shared__ unsigned before_s;
  if (a.cap < a.nwindows_cap) {
    const unsigned lin = sc.chunk_base + (tp.y0w * sc.nx + tp.x0w) / kChunkItems;
    if (tid < 64u) {
      const unsigned g1 = lin >> kLbpGroupShift, g2 = lin >> kLbpSuperShift;
      unsigned *hs = a.hits_super + (size_t)blockIdx.z * a.nsupers;
      unsigned *hg = a.hits_group + (size_t)blockIdx.z * a.ngroups;
      unsigned before = 0;
      for (unsigned q = tid; q < g2; q += 64u) before += atomicAdd(&hs[q], 0u);
      const unsigned gq = (g2 << (kLbpSuperShift - kLbpGroupShift)) + tid;
      if (gq < g1) before += atomicAdd(&hg[gq], 0u);
      before = wave_sum(before);
      if (tid == 0) before_s = before;
    }
    __syncthreads();
    if (before_s >= a.cap) return;
  }
  /* ---- LDS: cascade tables (geometry re-based to the tile's row stride), pair-phase lane layouts, truth tables, queues, tile */
  const unsigned step = (unsigned)a.step;
  const unsigned TS = (TW - 1u) * step + (unsigned)sc.win_w + 1u, TR = (TH - 1u) * step + (unsigned)sc.win_h + 1u;
  LbpLds t = lbp_stage_tables(smem, a, a.geom + (size_t)si * a.nweaks, tid, NT);
  char *extra = smem + ((lbp_lds_bytes(a.nstages, a.nweaks, a.nsub) + 15) & ~(size_t)15);
  LbpPairStage *pst = (LbpPairStage *)extra;
  extra += ((size_t)a.nstages * sizeof(LbpPairStage) + 15) & ~(size_t)15;
  uint32_t *truth = (uint32_t *)extra;
  extra += ((size_t)a.ntruth * 4 + 15) & ~(size_t)15;
  uint16_t *queue_all = (uint16_t *)extra; /* [NW][R * 64]: dword offsets of the survivors' top-left corners inside the tile */
  unsigned *tile = (unsigned *)(extra + (((size_t)TW * TH * 2 + 15) & ~(size_t)15));
  for (unsigned s = tid; s < a.nstages; s += NT) {
    const unsigned n = a.stage[s].count;
    LbpPairStage p;
    p.wn = (n >= 1u && n <= 32u) ? 64u / n : 0u;
    p.magic = p.wn ? 65536u / n + 1u : 0u;
    pst[s] = p;
  }
  for (unsigned i = tid; i < a.ntruth; i += NT) truth[i] = a.truth[i];
  __syncthreads(); /* lbp_stage_tables' copies are visible */
  /* the staged geometry holds byte offsets for the table's row stride a.S: re-base to TS (fy = off0 / (4 S), fx the rest);
   * the spare word takes the classifier's subset range (lbp_pair_hit) */
  for (unsigned i = tid; i < a.nweaks; i += NT) {
    LbpGeom *gp = (LbpGeom *)t.geom + i;
    const LbpGeom g = *gp;
    const LbpWeak wk = t.weak[i];
    const unsigned o = (unsigned)g.off0 >> 2, fy = o / a.S, fx = o - fy * a.S;
    LbpGeom r;
    r.off0 = (int)((fy * TS + fx) * 4u), r.fw = g.fw, r.fh_stride = (int)((unsigned)g.pad * TS * 4u);
    r.pad = (int)((wk.sub_off & 0xffffu) | ((wk.nsub < 8u ? wk.nsub : 8u) << 16)); /* a code has 8 bits: words >= 8 never match */
    *gp = r;
  }
  {
    const unsigned *Pg = a.padded + (size_t)blockIdx.z * a.frame_stride;
    const unsigned rows = (unsigned)(a.frame_stride / a.S);
    const unsigned gx0 = tp.x0w * step, gy0 = tp.y0w * step;
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
  __syncthreads();
  /* ---- from here on the waves are on their own */
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
    e = ph.end[0] < a.nstages ? ph.end[0] : a.nstages;
    for (;;) { /* wave-uniform */
#pragma clang loop unroll(disable) /* one copy of the stage loop, not R: registers (64 for the 1024-thread shapes) and code size */
      for (unsigned k = 0; k < R; k++) {
        if ((alive >> k) & 1u) {
          if (!lbp_window_stages<false, COUNT>(t, tile, odw_of(k) * 4u, 0u, s_prev, e, &evals)) alive &= ~(1u << k);
        }
      }
      const unsigned c = wave_sum((unsigned)__popc(alive));
      if (e >= a.nstages || c == 0u || c * 10u <= ph.adaptive_tenths * nvalid || e >= emax) {
        m = c;
        break;
      }
      s_prev = e, e = e + 1u;
    }
    if (m) {
      if (e >= a.nstages) { /* short cascade: the dense phase was all of it */
#pragma unroll
        for (unsigned k = 0; k < R; k++)
          if ((alive >> k) & 1u) lbp_tile_publish(a, sc, tp, odw_of(k), TS);
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
    const LbpPairStage ps = pst[s];
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
          pass = (truth[tt - 1u + (mine >> 5)] >> (mine & 31u)) & 1u;
        } else {
          float sum = 0.0f;
          for (unsigned i = 0; i < n; i++) { /* the reference's order of adds; the leaf values are wave-uniform reads */
            const LbpWeak wk = t.weak[first + i];
            sum += ((mine >> i) & 1u) ? wk.left : wk.right;
          }
          pass = !(sum < st.threshold);
        }
        const bool lead = live && k == 0u && pass;
        const uint64_t pm = ballot(lead);
        if (pm) {
          if (lastst) {
            if (lead) lbp_tile_publish(a, sc, tp, odw, TS);
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
        if (live) pass = lbp_window_stages<false, COUNT>(t, tile, odw * 4u, 0u, s, s + 1u, &evals);
        const uint64_t pm = ballot(pass);
        if (pm) {
          if (lastst) {
            if (pass) lbp_tile_publish(a, sc, tp, odw, TS);
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
