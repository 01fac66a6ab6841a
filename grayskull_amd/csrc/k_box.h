/*
 * k_box.h -- clipped box sums of any radius <= 127 without an integral image, for gs_blur with
 * radius > 3 (grayskull.h:268-283) and gs_adaptive_threshold (:230-247).
 *
 * The integral-image route (k_integral_* + k_box_px) moves ~16 B/px.  This one moves 3-4: a
 * 256-thread block owns a band of rows of one frame and spans the whole row (thread = 16 px,
 * w <= 4096).  VERTICAL FIRST (round 2): a thread keeps the column sums Vc of its 16 pixels over the
 * rows y-r .. y+r in registers (8 packed u16 pairs, <= 255 * 255) and slides them down one row per step,
 * Vc += row(y+r+1) - row(y-r): two 16-byte loads and byte unpacking, nobody else's data.  The box sums
 * of row y are then ONE horizontal pass over the block's Vc row, staged in LDS as u16 with a zero halo
 * of 128 entries either side (= the clipped sum): each thread adds up its first window and slides it
 * 15 times (two u16 reads per step at offsets that depend on the radius only and are computed once).
 * Round 1 kept a u32 vertical sum of horizontal sums instead and formed the horizontal sums of the
 * entering and of the leaving row every step: two LDS passes over byte rows per output row, their
 * offsets recomputed every time (435 VALU + 389 SALU instructions and 100 ds_read_u8 per 16-px row).
 * Every step reads two source rows (2 B/px; + the centre row for the adaptive compare) and writes one.
 *
 * Division by the number of in-image taps (cx * cy): estimate with two float multiplies by
 * precomputed reciprocals, then make it exact with the integer remainder (the estimate is within
 * +-1).  Sums stay below 2^24 (255 * 255^2).
 */
#ifndef GS_K_BOX_H
#define GS_K_BOX_H
#include "k_strip.h" /* static_for */

namespace gs {

/* LDS row layout (u16 entries): column x lives at BYTE kBoxPad + 2 x + 4 floor(x / 16): every thread's
 * 16-entry segment starts 36 bytes = 9 dwords after its neighbour's, so the reads of a wave (one per
 * lane, same k) fall into different banks (9 is odd); 32-byte stride would be a 4-way conflict. */
constexpr unsigned kBoxPad = 320, kBoxRowBytes = kBoxPad + 36 * 256 + kBoxPad;
GS_DEV int box_off(int x) { return 2 * x + 4 * (x >> 4); } /* arithmetic shift = floor for x < 0 */

/* floor(sum / (cx * cy)) for sum < 2^24 given rx ~ 1/cx, ry ~ 1/cy */
GS_DEV unsigned box_div(unsigned sum, unsigned cx, unsigned cy, float rx, float ry) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  const unsigned cnt = cx * cy;
  unsigned q = (unsigned)((float)sum * rx * ry);
  const int rem = (int)sum - (int)(q * cnt);
  if (rem < 0) q--;
  else if (rem >= (int)cnt) q++;
  return q;
}

/* MODE 0: dst = mean (gs_blur); MODE 1: dst = src > (int)(mean - (unsigned)c) ? 255 : 0.
 * grid (1, nbands, n frames), block 64 / 128 / 256 threads >= ceil(w / 16) (narrow frames: more blocks per CU instead of
 * idle waves); T rows per band; 1 <= r <= 127, 32 <= w <= 4096, any alignment.
 * Ragged rows (m = w % 16 != 0, round 4): the strips stay on the 16-px grid, the column sums of the last strip's
 * 16 - m columns past the row end are 0 like everything outside the image.  Its loads must not cross the row end
 * (behind the last row lies another frame, or nothing): it loads the row's last 16 bytes and shifts them down into
 * grid position, zeros entering; its m result bytes go out as 8 + 4 + 2 + 1-byte stores (no byte of the next row is
 * touched).  (Round 4, measured and not kept: loads from the dword-aligned address below a lane's pixels + v_alignbyte for
 * rows at byte phases that are no multiple of 4, the REALIGN recipe of k_strip.h -- 3838 x 2160, r = 5: 1.95 x the aligned
 * frame's time with and without it, profiles/r04p_box_realign_not_kept.log.  This kernel is issue-bound, and what ragged
 * rows cost it is the tail strip's extra instructions -- shift, four partial stores -- in ONE wave that the per-row barrier
 * makes the whole block wait for.) */
template <int MODE>
__global__ __launch_bounds__(256) void k_box16(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                                               unsigned T, size_t frame_bytes, unsigned r, int c) {
  __shared__ __attribute__((aligned(16))) uint8_t rows[2][kBoxRowBytes]; /* [phase]: the block's column sums as u16 */
  const unsigned tid = threadIdx.x, x0 = tid * 16u;
  const bool act = x0 < w;
  const unsigned m = w & 15u;                   /* block-uniform: bytes of the last strip inside the row */
  const bool tail = act && x0 + 16u > w;        /* m != 0 and this thread owns that strip */
  const uint32_t ld_off = !act ? kOOB : tail ? w - 16u : x0, st_off = (act && !tail) ? x0 : kOOB;
  const bool tail_wave = ballot(tail) != 0ull; /* wave-uniform */
  const BufRsrc S = make_buf(src + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const BufRsrc D = make_buf(dst + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const int y0 = (int)(blockIdx.y * T);
  if (y0 >= (int)h) return; /* whole block */
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  /* zero both row buffers once: the halos and the pad bytes after every 16 entries are never written again */
  for (unsigned i = tid; i < 2u * kBoxRowBytes / 4u; i += blockDim.x) ((uint32_t *)&rows[0][0])[i] = 0;
  /* per-pixel column counts and their reciprocals */
  unsigned cx[16];
  float rcx[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int x = (int)x0 + j;
    const int xa = x - (int)r < 0 ? 0 : x - (int)r, xb = x + (int)r > (int)w - 1 ? (int)w - 1 : x + (int)r;
    cx[j] = (act && x < (int)w) ? (unsigned)(xb - xa + 1) : 1u;
    rcx[j] = 1.0f / (float)cx[j];
  }
  /* MODE 0, rows whose window is not clipped vertically (cy = 2r + 1), r <= 31: the quotient is ONE v_mul_hi_u32 with
   * M = ceil(2^32 / cnt).  mul_hi(H, M) = floor(H / cnt + H (M cnt - 2^32) / (cnt 2^32)) and the excess is below
   * H / 2^32 <= 255 cnt / 2^32, which is less than the 1 / cnt that separates H / cnt from the next integer as long
   * as 255 cnt^2 < 2^32, i.e. cnt <= 4103 >= 63^2. */
  const bool magic_ok = MODE == 0 && r <= 31u;
  unsigned Mi[16];
#pragma unroll
  for (int j = 0; j < 16; j++) Mi[j] = MODE == 0 ? 0xffffffffu / (cx[j] * (2u * r + 1u)) + 1u : 0u; /* cnt >= 4 */
  auto row_load = [&](int yy) { /* this thread's 16 B of row yy, zeros outside the image */
    const U4 v = buf_load16(S, (yy >= 0 && yy < (int)h) ? (uint32_t)yy * w + ld_off : kOOB);
    uint32_t a = v.x, b = v.y, c = v.z, d = v.w; /* scalars: hipcc selects / merges whole structs through scratch memory */
    if (tail_wave) { /* wave-uniform: only the wave that holds the tail strip pays for the shift */
      const U4 sh = shift_down_bytes(v, 16u - m);
      a = tail ? sh.x : a, b = tail ? sh.y : b, c = tail ? sh.z : c, d = tail ? sh.w : d;
    }
    return U4{a, b, c, d};
  };
  /* column sums += / -= one row (bytes -> u16 pairs; fields cannot carry or borrow: 0 <= Vc <= 255 * 255) */
  uint32_t Vc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto vc_add = [&](const U4 &v) {
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) Vc[2 * q] += unpack_lo(d[q]), Vc[2 * q + 1] += unpack_hi(d[q]);
  };
  auto vc_sub = [&](const U4 &v) {
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) Vc[2 * q] -= unpack_lo(d[q]), Vc[2 * q + 1] -= unpack_hi(d[q]);
  };
  /* LDS byte offsets of the horizontal pass, relative to the thread's segment: they depend on r only
   * (wave-uniform, computed once): the first window's dword range and, for slide j, the entering entry
   * j + r and the leaving entry j - r - 1 */
  int ohi[16], olo[16];
#pragma unroll
  for (int j = 1; j < 16; j++) ohi[j] = (int)uniform((uint32_t)box_off(j + (int)r)), olo[j] = (int)uniform((uint32_t)box_off(j - (int)r - 1));
  const unsigned seg = kBoxPad + 36u * tid; /* = kBoxPad + box_off(x0) */
  auto hsum = [&](unsigned phase, unsigned (&H)[16]) {
    const uint8_t *p = &rows[phase][seg]; /* *(u16 *)(p + box_off(k)) = column sum at x0 + k, zero outside the image */
    /* first window x0-r .. x0+r: a single entry up to the next even k, then pairs (one dword = two entries, both
     * halves added by v_dot2 with ones; k even => 4-byte aligned, and a pair never straddles the 4 pad bytes
     * after 16 entries), then the last entry.  All bounds depend on r only: wave-uniform control flow. */
    unsigned s = 0;
    int k = -(int)r;
    if (k & 1) s += *(const uint16_t *)(p + box_off(k)), k++;
    for (; k + 1 <= (int)r; k += 2) s = udot2_ones(*(const uint32_t *)(p + box_off(k)), s);
    if (k <= (int)r) s += *(const uint16_t *)(p + box_off(k));
    H[0] = s;
#pragma unroll
    for (int j = 1; j < 16; j++) {
      s += *(const uint16_t *)(p + ohi[j]);
      s -= *(const uint16_t *)(p + olo[j]);
      H[j] = s;
    }
  };
  /* prologue: Vc = rows y0-r .. y0+r (no LDS, no barrier) */
  for (int yy = y0 - (int)r; yy <= y0 + (int)r; yy++) vc_add(row_load(yy)); /* block-uniform trip count */
  __syncthreads(); /* the zeroing above */
  U4 nin = row_load(y0 + (int)r + 1), nout = row_load(y0 - (int)r), ncen = MODE ? row_load(y0) : U4{0, 0, 0, 0};
  for (int i = 0; i < nrows; i++) { /* block-uniform */
    const int y = y0 + i;
    const unsigned ph = (unsigned)i & 1u;
    if (act) { /* this thread's 16 column sums of the window around row y: 32 bytes at a 4-byte aligned address */
      uint32_t *q = (uint32_t *)&rows[ph][seg];
#pragma unroll
      for (int k = 0; k < 8; k++) q[k] = Vc[k];
    }
    const U4 cen = ncen, in = nin, out = nout;
    __syncthreads(); /* also orders this phase's writes after the reads of two iterations ago */
    nin = row_load(y + (int)r + 2), nout = row_load(y - (int)r + 1);
    if (MODE) ncen = row_load(y + 1);
    unsigned H[16];
    hsum(ph, H);
    vc_add(in), vc_sub(out); /* the window around row y + 1 */
    const int ya = y - (int)r < 0 ? 0 : y - (int)r, yb = y + (int)r > (int)h - 1 ? (int)h - 1 : y + (int)r;
    const unsigned cy = (unsigned)(yb - ya + 1);
    const float rcy = 1.0f / (float)cy;
    uint32_t od[4] = {0, 0, 0, 0};
    const uint32_t cd[4] = {cen.x, cen.y, cen.z, cen.w};
    /* MODE 1 without the division: px > floor(V / cnt) - c  <=>  px + c >= floor(V / cnt) + 1  <=>  (px + c) * cnt > V
     * (cnt > 0).  k = px + c <= 0 can never exceed V >= 0, k >= 256 always does (V <= 255 cnt), in between the
     * product fits 32 bits (cnt <= 255^2).  Only for |c| < 2^30: beyond that the reference's unsigned `mean - c`
     * wraps and the literal form below reproduces it. */
    const bool by_product = MODE == 1 && c > -(1 << 30) && c < (1 << 30);
    const bool by_magic = magic_ok && cy == 2u * r + 1u; /* block-uniform */
#pragma unroll
    for (int j = 0; j < 16; j++) {
      unsigned o;
      if (MODE == 0 && by_magic) {
        o = __umulhi(H[j], Mi[j]) & 0xffu;
      } else if (MODE == 1 && by_product) {
        const int k = (int)((cd[j >> 2] >> (8 * (j & 3))) & 0xffu) + c;
        const unsigned kc = (unsigned)(k < 0 ? 0 : k > 256 ? 256 : k);
        o = kc * (cx[j] * cy) > H[j] ? 255u : 0u;
      } else {
        const unsigned q = box_div(H[j], cx[j], cy, rcx[j], rcy);
        if (MODE == 0) o = q & 0xffu;
        else {
          const int thr = (int)(q - (unsigned)c);
          const int px = (int)((cd[j >> 2] >> (8 * (j & 3))) & 0xffu);
          o = px > thr ? 255u : 0u;
        }
      }
      od[j >> 2] |= o << (8 * (j & 3));
    }
    buf_store16(D, (uint32_t)y * w + st_off, U4{od[0], od[1], od[2], od[3]}); /* st_off = kOOB: dropped */
    if (tail_wave) buf_store_first(D, tail ? (uint32_t)y * w + x0 : kOOB, od[0], od[1], od[2], od[3], m);
  }
}

/* ------------------------------------------------------------------ radius known at compile time, r <= 8 */
/* k_box16 re-reads the row that leaves the window (and, for the adaptive compare, the centre row): it moves 3.1 / 4.2 B/px
 * where 2 are needed, and since round 2 removed most of its arithmetic that traffic IS its time (64 x 4K, r = 8: 1.66 GB per
 * launch at the 5.3 TB/s two-buffer rate = 310 of the 339 us measured, VALU issue ~45 % busy; profiles/r03z_pmc_box.txt).
 * Here the 2 RR + 1 raw rows of the window stay in registers (4 VGPRs each; the row loop is unrolled 2 RR + 1 times so the
 * ring is indexed statically): every source row is loaded once per band, the leaving and the centre row come from the
 * ring.  With the radius a constant the LDS offsets of the horizontal pass are immediates, the tap counts are constants
 * (only the block's first / last thread owns horizontally clipped pixels: w >= 32), the MODE 1 count (cx * cy) is a scalar
 * product and the MODE 0 quotient is v_mul_hi_u32 by ceil(2^32 / (cx * cy)) from a (RR + 1)^2 table (exact: 255 cnt^2 < 2^32,
 * see k_box16) -- for every row, clipped or not (h >= 2 RR + 1).  Same results as k_box16 (tests run both).
 * MODE 1 needs |c| < 2^30 (the launcher sends the wrap-around cases to k_box16). */
template <int RR> struct BoxMagic {
  uint32_t m[RR + 1][RR + 1]; /* [cy - (RR + 1)][cx - (RR + 1)] = ceil(2^32 / (cx * cy)) */
  constexpr BoxMagic() : m() {
    for (int a = 0; a <= RR; a++)
      for (int b = 0; b <= RR; b++) m[a][b] = 0xffffffffu / (uint32_t)((RR + 1 + a) * (RR + 1 + b)) + 1u;
  }
};
#ifndef GS_BOXR_ATTR
#define GS_BOXR_ATTR
#endif
template <int MODE, int RR>
__global__ __launch_bounds__(256) GS_BOXR_ATTR void k_box16r(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned T,
                                                size_t frame_bytes, int c) {
  constexpr int N = 2 * RR + 1;
  static_assert(RR >= 1 && RR <= 16, "horizontally clipped pixels only in the block's first / last thread (w >= 32)");
  static constexpr BoxMagic<RR> kMagic{};
  __shared__ __attribute__((aligned(16))) uint8_t rows[2][kBoxRowBytes];
  const unsigned tid = threadIdx.x, x0 = tid * 16u;
  /* Ragged rows (w % 16 != 0, round 5): this kernel is the BODY -- the whole strips only, and the last of them is a feeder:
   * its column sums enter the LDS row for its left neighbour's windows, its results (clipped at the wrong column) are
   * dropped.  No pixel left of it is within RR <= 16 columns of the row's end, so nothing here depends on w % 16;
   * k_box_edge writes columns (w & ~15) - 16 .. w - 1 in a launch of its own. */
  const unsigned wb = w & ~15u;
  const bool feeder = wb != w && x0 + 16u == wb;
  const bool act = x0 < wb, first = x0 == 0, last = x0 + 16u == w, keep = act && !feeder;
  const BufRsrc S = make_buf(src + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const BufRsrc D = make_buf(dst + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const int y0 = (int)(blockIdx.y * T);
  if (y0 >= (int)h) return; /* whole block */
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  for (unsigned i = tid; i < 2u * kBoxRowBytes / 4u; i += blockDim.x) ((uint32_t *)&rows[0][0])[i] = 0;
  auto row_load = [&](int yy) {
    return buf_load16(S, (act && yy >= 0 && yy < (int)h) ? (uint32_t)yy * w + x0 : kOOB);
  };
  uint32_t Vc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto vc_add = [&](const U4 &v) {
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) Vc[2 * q] += unpack_lo(d[q]), Vc[2 * q + 1] += unpack_hi(d[q]);
  };
  auto vc_sub = [&](const U4 &v) {
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) Vc[2 * q] -= unpack_lo(d[q]), Vc[2 * q + 1] -= unpack_hi(d[q]);
  };
  const unsigned seg = kBoxPad + 36u * tid;
  auto hsum = [&](unsigned phase, unsigned (&H)[16]) { /* as in k_box16, every offset a constant */
    const uint8_t *p = &rows[phase][seg];
    unsigned s = 0;
    int k = -RR;
    if (k & 1) s += *(const uint16_t *)(p + box_off(k)), k++;
#pragma unroll
    for (; k + 1 <= RR; k += 2) s = udot2_ones(*(const uint32_t *)(p + box_off(k)), s);
    if (k <= RR) s += *(const uint16_t *)(p + box_off(k));
    H[0] = s;
#pragma unroll
    for (int j = 1; j < 16; j++) {
      s += *(const uint16_t *)(p + box_off(j + RR));
      s -= *(const uint16_t *)(p + box_off(j - RR - 1));
      H[j] = s;
    }
  };
  /* prologue: the window of row y0 into the ring (slot k = row y0 - RR + k) and into the column sums */
  U4 ring[N];
  static_for<N>([&](auto K) { ring[decltype(K)::value] = row_load(y0 - RR + decltype(K)::value); });
  static_for<N>([&](auto K) {
    constexpr int k = decltype(K)::value;
    vc_add(ring[k]);
  });
  __syncthreads(); /* the zeroing above */
  U4 nin = row_load(y0 + RR + 1);
  for (int base = 0; base < nrows; base += N) {
    static_for<N>([&](auto I_) {
      constexpr int I = decltype(I_)::value; /* slot of the row that leaves after this step: row y - RR */
      const int i = base + I;
      if (i >= nrows) return; /* block-uniform */
      const int y = y0 + i;
      const unsigned ph = (unsigned)i & 1u;
      if (act) {
        uint32_t *q = (uint32_t *)&rows[ph][seg];
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = Vc[k];
      }
      const U4 in = nin;
      sched_fence();
      __syncthreads(); /* also orders this phase's writes after the reads of two iterations ago */
      sched_fence(); /* rows do not mix: the ring stays packed, 4 registers per row */
      nin = row_load(y + RR + 2);
      unsigned H[16];
      hsum(ph, H);
      const U4 cen = ring[(I + RR) % N]; /* row y */
      sched_fence(); /* the other rows of the ring stay packed: nothing of theirs is unpacked ahead of time */
      vc_add(in), vc_sub(ring[I]); /* the window around row y + 1 */
      sched_fence();
      ring[I] = U4{opaque(in.x), opaque(in.y), opaque(in.z), opaque(in.w)}; /* 4 registers, not the 8 unpacked ones */
      const int ya = y - RR < 0 ? 0 : y - RR, yb = y + RR > (int)h - 1 ? (int)h - 1 : y + RR;
      const unsigned cy = (unsigned)(yb - ya + 1); /* RR + 1 .. N, block-uniform */
      uint32_t od[4] = {0, 0, 0, 0};
      const uint32_t cd[4] = {cen.x, cen.y, cen.z, cen.w};
#pragma unroll
      for (int j = 0; j < 16; j++) {
        /* tap columns of pixel j: N, unless the block's first thread clips it on the left (j < RR: RR + 1 + j columns) or
         * its last thread on the right (j >= 16 - RR: RR + 16 - j); constants once unrolled */
        const int el = j < RR ? j : RR, er = j >= 16 - RR ? 15 - j : RR;
        const bool cl = j < RR && first, cr = j >= 16 - RR && last;
        unsigned o;
        if (MODE == 0) {
          const unsigned a = cy - (unsigned)(RR + 1);
          /* uniform(): the three multipliers are scalars picked per lane, not a load from a per-lane address */
          const uint32_t ml = uniform(kMagic.m[a][el]), mr = uniform(kMagic.m[a][er]), mc = uniform(kMagic.m[a][RR]);
          o = __umulhi(H[j], cl ? ml : cr ? mr : mc) & 0xffu;
        } else {
          const unsigned cc = (unsigned)N * cy, ccl = (unsigned)(RR + 1 + el) * cy, ccr = (unsigned)(RR + 1 + er) * cy; /* scalar */
          const int k = (int)((cd[j >> 2] >> (8 * (j & 3))) & 0xffu) + c;
          const unsigned kc = (unsigned)(k < 0 ? 0 : k > 256 ? 256 : k);
          o = kc * (cl ? ccl : cr ? ccr : cc) > H[j] ? 255u : 0u;
        }
        od[j >> 2] |= o << (8 * (j & 3));
      }
      buf_store16(D, keep ? (uint32_t)y * w + x0 : kOOB, U4{od[0], od[1], od[2], od[3]});
    });
  }
}

/* ------------------------------------------------------------------ the ragged edge of k_box16r's frames */
/* Columns (w & ~15) - 16 .. w - 1 of a frame whose rows are no multiple of 16 long (the last whole strip, which the body
 * launch only feeds on, and the m = w % 16 columns behind it): one WAVE per band of rows, lane l <-> column
 * (w & ~15) - 16 - r + l, so the 16 + m output columns and the r columns either side of them are all in the wave
 * (16 + m + 2 r <= 63 for r <= 16).  Vertical first like the body: a lane slides the sum of its column over rows
 * y - r .. y + r (two byte loads per row, one step ahead); the window sums are then two reads of the wave's inclusive scan.
 * 0.8 % of a 4K frame's pixels, in a launch of its own so that no block of the body waits at its per-row barrier for a
 * tail strip's shifts and partial stores (rounds 3-4: k_box16's ragged rows, 1.3 x the aligned time; a ragged form of
 * the ring kernel, no faster than that).  Divisions as in k_box16. */
template <int MODE>
__global__ __launch_bounds__(64) void k_box_edge(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned T,
                                                 size_t frame_bytes, unsigned r, int c) {
  const unsigned lane = threadIdx.x;
  const int xo0 = (int)(w & ~15u) - 16, x = xo0 - (int)r + (int)lane;
  const bool col = x >= 0 && x < (int)w, outl = x >= xo0 && x < (int)w;
  const BufRsrc S = make_buf(src + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const BufRsrc D = make_buf(dst + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const int y0 = (int)(blockIdx.y * T);
  if (y0 >= (int)h) return; /* whole wave */
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  /* every load and store is issued unconditionally (rows / columns outside the image: the buffer's out-of-range offset,
   * which reads 0 and drops writes), so hipcc can count them: with loads inside branches it waited for ALL outstanding
   * memory operations -- the previous rows' stores included -- at the top of every trip */
  auto ld = [&](int yy) -> unsigned { return buf_load1(S, (col && yy >= 0 && yy < (int)h) ? (uint32_t)yy * w + (uint32_t)x : kOOB); };
  unsigned Vc = 0;
  for (int yy = y0 - (int)r; yy <= y0 + (int)r; yy += 8) { /* wave-uniform trip count; eight loads in flight (one at a time:
                                                            * 2 r + 1 memory latencies in a row before the first result) */
    unsigned t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = ld(yy + j <= y0 + (int)r ? yy + j : -1);
#pragma unroll
    for (int j = 0; j < 8; j++) Vc += t[j];
  }
  const int xa = x - (int)r < 0 ? 0 : x - (int)r, xb = x + (int)r > (int)w - 1 ? (int)w - 1 : x + (int)r;
  const unsigned cx = outl ? (unsigned)(xb - xa + 1) : 1u;
  const float rcx = 1.0f / (float)cx;
  const unsigned full = 2u * r + 1u, cnt_full = cx * full, Mi = 0xffffffffu / cnt_full + 1u;
  const unsigned l_hi = lane + r < 63u ? lane + r : 63u, l_lo = lane > r ? lane - r - 1u : 0u;
  /* A wave's rows are 0.5-1 us of memory latency apart (every row of the strip is a line of its own) and depend on each
   * other through Vc: the loads run U rows ahead (first form, one row ahead and the start-up loads one at a time: 41 / 61 us
   * per 64 x 3838x2160 for blur / adaptive -- a fifth of the body launch's time for 0.8 % of its pixels;
   * profiles/r05l_box_edge_one_row_ahead.log). */
  constexpr int U = 8;
  unsigned nin[U], nout[U], ncen[U];
#pragma unroll
  for (int j = 0; j < U; j++) nin[j] = ld(y0 + (int)r + 1 + j), nout[j] = ld(y0 - (int)r + j), ncen[j] = MODE ? ld(y0 + j) : 0u;
  const bool by_product = MODE == 1 && c > -(1 << 30) && c < (1 << 30);
  for (int i0 = 0; i0 < nrows; i0 += U) { /* wave-uniform */
    unsigned in[U], out[U], cen[U];
#pragma unroll
    for (int j = 0; j < U; j++) in[j] = nin[j], out[j] = nout[j], cen[j] = ncen[j];
#pragma unroll
    for (int j = 0; j < U; j++) { /* the next trip's rows (past the band's end: loaded for nothing, at most once per band) */
      const int yn = y0 + i0 + U + j;
      nin[j] = ld(yn + (int)r + 1), nout[j] = ld(yn - (int)r);
      if (MODE) ncen[j] = ld(yn);
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int y = y0 + i0 + j;
      const unsigned P = wave_incl_scan(Vc);
      const unsigned p_hi = shfl(P, (int)l_hi), p_lo = shfl(P, (int)l_lo);
      const unsigned H = p_hi - (lane > r ? p_lo : 0u);
      Vc += in[j] - out[j];
      const int ya = y - (int)r < 0 ? 0 : y - (int)r, yb = y + (int)r > (int)h - 1 ? (int)h - 1 : y + (int)r;
      const unsigned cy = yb >= ya ? (unsigned)(yb - ya + 1) : 1u; /* rows past the image's end (never stored): any divisor */
      unsigned o;
      if (cy == full && (MODE == 0 || by_product)) { /* wave-uniform: the rows whose window is not clipped vertically */
        if (MODE == 0) {
          o = __umulhi(H, Mi) & 0xffu; /* exact for cnt <= 4103 (k_box16); here cnt <= 33 * 33 */
        } else {
          const int k = (int)cen[j] + c;
          const unsigned kc = (unsigned)(k < 0 ? 0 : k > 256 ? 256 : k);
          o = __umul24(kc, cnt_full) > H ? 255u : 0u; /* both factors below 2^24 */
        }
      } else if (MODE == 1 && by_product) { /* see k_box16 */
        const int k = (int)cen[j] + c;
        const unsigned kc = (unsigned)(k < 0 ? 0 : k > 256 ? 256 : k);
        o = kc * (cx * cy) > H ? 255u : 0u;
      } else {
        const unsigned q = box_div(H, cx, cy, rcx, 1.0f / (float)cy);
        if (MODE == 0) o = q & 0xffu;
        else o = (int)cen[j] > (int)(q - (unsigned)c) ? 255u : 0u;
      }
      buf_store1(D, (outl && i0 + j < nrows) ? (uint32_t)y * w + (uint32_t)x : kOOB, o);
    }
  }
}

}  // namespace gs
#endif
