/*
 * k_lbp_dense.h -- the first stages of gs_lbp_window (grayskull.h:790-813) for EVERY window of a frame,
 * with the integral-image rows shared between vertically neighbouring windows.
 *
 * Why: the cascade is bound by the texture path's gather rate (profiles/r01c_lbp_pmc.txt: TA busy 94 %,
 * ~7 CU-cycles per wave-wide dword gather whatever the phase), and 7 of the ~10.4 classifier evaluations a
 * wave performs per 64 windows belong to stages 0-1, which k_lbp_cascade runs one window per lane with 16
 * gathers per weak classifier.  A weak classifier reads a 4 x 4 grid of table corners with row pitch fh:
 * the windows y, y + fh, y + 2 fh, ... of one column share three of their four corner rows.  So here a wave
 * owns a TILE of 64 window columns (lane = column) x 64 window rows and walks every residue class
 * r = 0 .. fh-1 of rows downwards: ONE new corner row (4 gathers of 64 consecutive dwords, the cheapest
 * pattern) per window instead of four, the previous rows and the 3 x 3 cell sums of the two cell rows above
 * stay in registers (rotating by renaming: the walk is unrolled by three).  Gathers per window and weak
 * classifier: 4 (64 + 3 fh) / 64 = 4.6 (fh = 3) .. 7 (fh = 16) instead of 16; VALU work per window drops too
 * (6 subtractions for the new cell row instead of 21 for nine cells).
 *
 * Every window of the tile is evaluated (dead ones included: their rows are needed anyway), so a weak
 * classifier's result is one BIT per window, kept as a 64-bit row mask per lane, and the stage decision
 * `sum < threshold` (sequential float32 adds in weak order, ref :796-810) becomes a truth table over the
 * stage's match bits -- computed on the host with exactly those adds (gs_api.cpp: pass_lut) and applied to
 * 64 windows at once with bitwise operations.  Stages of up to 5 weak classifiers qualify.
 *
 * Output: one bit per window (alive after stages [0, pre_stages)) in window-row-major words, which
 * k_lbp_cascade then takes as its starting set instead of "all windows" (a.pre_stages): ordering, the
 * max_rects exit and the later stages are untouched.  Requirements (checked by the launcher): step == 1,
 * no GUARD geometry, every prefiltered stage has <= 5 weak classifiers.
 */
#ifndef GS_K_LBP_DENSE_H
#define GS_K_LBP_DENSE_H
#include <utility>

#include "k_lbp.h"

namespace gs {

constexpr unsigned kPreTile = 64;     /* window rows per tile = bits of a lane's row mask */
constexpr unsigned kPreMaxWeaks = 5;  /* weak classifiers per prefiltered stage (32-entry truth table) */

struct LbpPreArgs {
  const unsigned *pass_lut;     /* per stage: bit b set <=> the stage passes when weak k matched iff bit k of b */
  unsigned long long *bitmap;   /* n frames x a.pre_words (written here, read by k_lbp_cascade) */
  const unsigned *not_integral; /* n frames: != 0 when the frame's table is not an integral image of bytes */
  unsigned small_cells;         /* every cell of every prefiltered classifier covers < 2^31 / 255 pixels */
  unsigned xcd_swizzle;
};

/* Rows in flight ahead of the arithmetic.  The table of a 4K frame (33 MB) does not stay in an XCD's 4 MB L2 while
 * ~1000 tiles per XCD walk it, so a row comes from the Infinity Cache / HBM (~0.7 us): with one row ahead the
 * kernel sat on that latency (204 SIMD-cycles per step against ~96 of VALU work, first measurement of round 3). */
#ifndef GS_LBP_DENSE_PD
#define GS_LBP_DENSE_PD 2
#endif
constexpr int kPreRing = GS_LBP_DENSE_PD + 2; /* raw rows t-1, t, t+1 .. t+PD; the walk is unrolled by this */
struct LbpRowRing { unsigned G[kPreRing][4], R[kPreRing][3]; };

/* the 4 corners of one table row of this residue class: 64 consecutive dwords per gather.  Buffer loads: the
 * lane's constant column offset is the VGPR offset, everything else -- tile row, residue class, feature offset,
 * step x row pitch, corner x cell width -- is one wave-uniform SGPR offset, so a gather costs NO vector ALU
 * (global_load with a per-lane 64-bit address took v_mad_u64_u32 + 3 v_lshl_add_u64 per row).
 * GS_LBP_DENSE_BUF=0 keeps the global-load form for A/B runs. */
#ifndef GS_LBP_DENSE_BUF
#define GS_LBP_DENSE_BUF 1
#endif
struct LbpTable {
  BufRsrc buf;     /* one frame's padded table */
  const char *Pg;  /* the same, as a pointer */
};
template <int SLOT>
GS_DEV void lbp_dense_load(LbpRowRing &q, const LbpTable &T, unsigned base, unsigned tstep, unsigned last, unsigned fhS,
                           unsigned colB, unsigned fwB) {
  const unsigned tt = tstep < last ? tstep : last; /* steps past the class's last row re-read it (never used) */
  const unsigned row = base + tt * fhS;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#if GS_LBP_DENSE_BUF
    q.G[SLOT][i] = buf_gather4(T.buf, colB, row + (unsigned)i * fwB);
#else
    q.G[SLOT][i] = *(const unsigned *)((T.Pg + (size_t)(row + (unsigned)i * fwB)) + colB);
#endif
  }
}

/* (cell >= centre) for the eight neighbours, most significant first (tl tc tr r br bc bl l, ref :780-782).
 * GS_LBP_SIGNED_CELLS: every cell here is a true box sum of <= 2^31 / 255 pixels (the launcher checks the
 * geometry, k_integral_pad checks that the table is an integral image of bytes), so `cell < centre` is the sign
 * bit of cell - centre: one subtraction + one v_alignbit_b32 per neighbour shifts it into the code (no VCC
 * round trip: v_cmp -> v_cndmask costs two wait states on gfx950). */
GS_DEV unsigned lbp_code8(const unsigned (&n)[8], unsigned ctr, bool small_cells) {
  if (small_cells) {
    unsigned lt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) lt = alignbit(lt, n[k] - ctr, 31); /* (lt << 1) | sign(n[k] - ctr) */
    return lt ^ 255u;
  }
  unsigned code = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) code |= (unsigned)(n[k] >= ctr) << (7 - k);
  return code;
}

/* one step of the walk: row `tstep` is in slot NEW, row tstep-1 in PRV; prefetch row tstep+PD into the slot row
 * tstep-2 just left, form cell row tstep-1, and evaluate window tstep-3 of the class from cell rows tstep-3,
 * tstep-2, tstep-1 */
template <int U, bool SMALL>
GS_DEV void lbp_dense_step(LbpRowRing &q, const LbpLds &t, const LbpTable &T, unsigned base, unsigned tstep, unsigned last,
                           unsigned nwin, unsigned fhS, unsigned colB, unsigned fwB, unsigned r, unsigned fh,
                           unsigned sub_off, unsigned nsub, uint64_t rowany, uint64_t &bits) {
  /* step t = 1 + kPreRing * q + U: raw row t in slot t % ring, cell row c in slot c % ring */
  constexpr int NG = kPreRing, NEW = (1 + U) % NG, PRV = (NEW + NG - 1) % NG, FAR = (NEW + GS_LBP_DENSE_PD) % NG;
  constexpr int TOP = (NEW + NG - 3) % NG, MID = (NEW + NG - 2) % NG; /* cell rows t-3, t-2; t-1 goes to PRV */
  lbp_dense_load<FAR>(q, T, base, tstep + GS_LBP_DENSE_PD, last, fhS, colB, fwB);
  unsigned V[4]; /* vertical differences first: 4 + 3 subtractions per cell row */
#pragma unroll
  for (int i = 0; i < 4; i++) V[i] = q.G[NEW][i] - q.G[PRV][i];
#pragma unroll
  for (int i = 0; i < 3; i++) q.R[PRV][i] = V[i + 1] - V[i];
  if (tstep >= 3u && tstep - 3u < nwin) { /* wave-uniform */
    const unsigned ybit = r + (tstep - 3u) * fh;
    if ((rowany >> ybit) & 1ull) { /* some lane's window of this row is still alive */
      const unsigned nb[8] = {q.R[TOP][0], q.R[TOP][1], q.R[TOP][2], q.R[MID][2],
                              q.R[PRV][2], q.R[PRV][1], q.R[PRV][0], q.R[MID][0]};
      const unsigned code = lbp_code8(nb, q.R[MID][1], SMALL);
      /* subset bit, branch-free: words past the classifier's subset count read word 0 and are masked */
      const unsigned word = code >> 5;
      const unsigned v = (uint32_t)t.subsets[sub_off + (word < nsub ? word : 0u)];
      const unsigned hit = (word < nsub ? 1u : 0u) & (v >> (code & 31u)); /* bit 0 only */
      bits |= (uint64_t)hit << ybit;
    }
  }
}

template <int... I>
GS_DEV void lbp_dense_prologue(LbpRowRing &q, const LbpTable &T, unsigned base, unsigned last, unsigned fhS, unsigned colB,
                               unsigned fwB, std::integer_sequence<int, I...>) {
  (lbp_dense_load<I>(q, T, base, (unsigned)I, last, fhS, colB, fwB), ...); /* rows 0 .. PD */
}
template <bool SMALL, int... U>
GS_DEV void lbp_dense_steps(LbpRowRing &q, const LbpLds &t, const LbpTable &T, unsigned base, unsigned tstep, unsigned last,
                            unsigned nwin, unsigned fhS, unsigned colB, unsigned fwB, unsigned r, unsigned fh,
                            unsigned sub_off, unsigned nsub, uint64_t rowany, uint64_t &bits, std::integer_sequence<int, U...>) {
  (lbp_dense_step<U, SMALL>(q, t, T, base, tstep + (unsigned)U, last, nwin, fhS, colB, fwB, r, fh, sub_off, nsub, rowany, bits), ...);
}

/* weak classifier wi for every window of the tile: bit y of the result = match of window row y0 + y */
template <bool SMALL>
GS_DEV uint64_t lbp_dense_weak(const LbpLds &t, const LbpTable &T, unsigned wi, unsigned colB, unsigned rowB,
                               unsigned tile_rowB, unsigned nrows, uint64_t rowany, unsigned *loads) {
  const LbpGeom g = t.geom[wi];
  const unsigned off0 = uniform((unsigned)g.off0), fwB = uniform((unsigned)g.fw), fhS = uniform((unsigned)g.fh_stride),
                 fh = uniform((unsigned)g.pad);
  const LbpWeak wk = t.weak[wi];
  const unsigned sub_off = uniform(wk.sub_off), nsub = uniform(wk.nsub);
  uint64_t bits = 0;
  const unsigned nclass = fh < nrows ? fh : nrows;
  for (unsigned r = 0; r < nclass; r++) {
    const unsigned nwin = (nrows - r + fh - 1u) / fh; /* windows y = r, r + fh, ... < nrows */
    const unsigned last = nwin + 2u;                  /* they need table rows 0 .. nwin + 2 of the class */
    const unsigned base = tile_rowB + r * rowB + off0; /* byte offset in the frame's table (< 2 GiB, launcher) */
    LbpRowRing q;
    lbp_dense_prologue(q, T, base, last, fhS, colB, fwB, std::make_integer_sequence<int, GS_LBP_DENSE_PD + 1>());
    for (unsigned tstep = 1u; tstep <= last; tstep += (unsigned)kPreRing)
      lbp_dense_steps<SMALL>(q, t, T, base, tstep, last, nwin, fhS, colB, fwB, r, fh, sub_off, nsub, rowany, bits,
                             std::make_integer_sequence<int, kPreRing>());
    if (loads) *loads += 4u * ((unsigned)GS_LBP_DENSE_PD + 1u + (unsigned)kPreRing * ((last + (unsigned)kPreRing - 1u) / (unsigned)kPreRing));
  }
  return bits;
}

/* stages [0, a.pre_stages) for the tile; returns the lane's row mask of surviving windows */
template <bool COUNT, bool SMALL>
GS_DEV uint64_t lbp_dense_tile(const LbpArgs &a, const LbpPreArgs &p, const LbpLds &t, const LbpTable &Pg, unsigned colB,
                               unsigned rowB, unsigned y0, unsigned nrows, bool lane_ok, unsigned &evals, unsigned &loads) {
  const uint64_t allrows = nrows >= 64u ? ~0ull : ((1ull << nrows) - 1ull);
  uint64_t alive = lane_ok ? allrows : 0ull;
  uint64_t rowany = allrows; /* rows in which some lane is alive (wave-uniform) */
  for (unsigned s = 0; s < a.pre_stages; s++) {
    const LbpStage st = t.stage[s];
    const unsigned first = uniform(st.first), count = uniform(st.count);
    uint64_t m[kPreMaxWeaks];
#pragma unroll
    for (unsigned k = 0; k < kPreMaxWeaks; k++) {
      m[k] = 0;
      if (k < count)
        m[k] = lbp_dense_weak<SMALL>(t, Pg, first + k, colB, rowB, y0 * rowB, nrows, rowany, COUNT ? &loads : nullptr);
    }
    /* stage decision for 64 windows at once: OR of the passing rows of the truth table */
    const unsigned lut = uniform(p.pass_lut[s]);
    uint64_t pass = 0;
    for (unsigned b = 0; b < (1u << count); b++) {
      if (!((lut >> b) & 1u)) continue;
      uint64_t tt = ~0ull;
#pragma unroll
      for (unsigned k = 0; k < kPreMaxWeaks; k++)
        if (k < count) tt &= ((b >> k) & 1u) ? m[k] : ~m[k];
      pass |= tt;
    }
    if constexpr (COUNT) evals += (unsigned)__popcll(alive) * count;
    alive &= pass;
    if (s + 1u < a.pre_stages) { /* rows that still matter to the next stage */
      uint64_t ra = 0;
      for (unsigned y = 0; y < nrows; y++)
        if (ballot((alive >> y) & 1ull)) ra |= 1ull << y;
      rowany = ra;
    }
  }
  return alive;
}

/* grid (ceil(max tiles per scale / 4) [rounded up to 8 with the XCD mapping], nscales of the group, n frames),
 * block 256 = 4 waves = 4 tiles consecutive in x; dynamic LDS = lbp_lds_bytes(...) + 16 */
template <bool COUNT = false>
__global__ __launch_bounds__(256) void k_lbp_dense(LbpArgs a, LbpPreArgs p) {
  GS_DYN_LDS(smem);
  const unsigned si = blockIdx.y + a.scale0;
  const LbpScale sc = a.scales[si];
  const LbpPreScale ps = a.pre_scales[si];
  const unsigned nblk = (ps.ntiles + 3u) >> 2;
  unsigned bx = blockIdx.x;
  if (p.xcd_swizzle) { /* XCD k takes the k-th eighth of the scale's tile rows (see k_lbp_cascade) */
    const unsigned per = (nblk + 7u) >> 3, j = blockIdx.x >> 3;
    if (j >= per) return;
    bx = (blockIdx.x & 7u) * per + j;
  }
  if (bx >= nblk) return; /* whole block */
  /* the scales issued before this launch already hold max_rects detections: nothing of this one can be among
   * the first max_rects (ref :819-823), and every chunk of it will skip too (k_lbp_cascade: before_s >= cap) */
  if (a.hits_total[blockIdx.z] >= a.cap) return;
  const unsigned tid = threadIdx.x;
  const LbpLds t = lbp_stage_tables(smem, a, a.geom + (size_t)si * a.nweaks, tid, 256u);
  __syncthreads();
  const unsigned lane = tid & 63u, tile = uniform(bx * 4u + (tid >> 6));
  if (tile >= ps.ntiles) return; /* whole wave; no block barrier follows */
  const unsigned ty = tile / ps.tiles_x, tx = tile - ty * ps.tiles_x;
  const unsigned x0 = tx * 64u, y0 = ty * kPreTile;
  const unsigned nrows = sc.ny - y0 < kPreTile ? sc.ny - y0 : kPreTile;
  const bool lane_ok = x0 + lane < sc.nx;
  const unsigned colB = (lane_ok ? x0 + lane : sc.nx - 1u) * 4u; /* idle lanes repeat the last column */
  const unsigned rowB = a.S * 4u;
  LbpTable Pg;
  Pg.Pg = (const char *)(a.padded + (size_t)blockIdx.z * a.frame_stride);
  Pg.buf = make_buf(Pg.Pg, a.frame_stride * 4u);
  unsigned evals = 0, loads = 0;
  /* sign-bit compares need cells that are true sums of < 2^31 / 255 bytes: geometry (p.small_cells, host) and a
   * table that really is an integral image of bytes (p.not_integral[frame], set by k_integral_pad otherwise) */
  const bool small = p.small_cells && !uniform(p.not_integral[blockIdx.z]);
  const uint64_t alive = small ? lbp_dense_tile<COUNT, true>(a, p, t, Pg, colB, rowB, y0, nrows, lane_ok, evals, loads)
                               : lbp_dense_tile<COUNT, false>(a, p, t, Pg, colB, rowB, y0, nrows, lane_ok, evals, loads);
  /* transpose: lane y collects the 64 column bits of window row y0 + y */
  uint64_t mine = 0;
  for (unsigned y = 0; y < nrows; y++) {
    const uint64_t w = ballot((alive >> y) & 1ull);
    if (lane == y) mine = w;
  }
  unsigned long long *bm = p.bitmap + (size_t)blockIdx.z * a.pre_words + ps.word_base;
  if (lane < nrows) bm[(size_t)(y0 + lane) * ps.wpr + tx] = mine;
  if constexpr (COUNT) {
    const unsigned ev = wave_sum(evals), ld = wave_sum(lane_ok ? loads : 0u);
    const unsigned nw = wave_sum(lane_ok ? nrows : 0u);
    if (lane == 0) {
      atomicAdd(a.evaluated + 3, (unsigned long long)nw);
      atomicAdd(a.evaluated + 1, (unsigned long long)ev);
      atomicAdd(a.evaluated + 2, (unsigned long long)ld);
    }
  }
}

}  // namespace gs
#endif
